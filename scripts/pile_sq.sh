#!/bin/bash
# Pile-up regime: SQ counters of the tick's kernels over the last 50 ticks of a 450-tick run (how busy the vector
# ALUs are in the search and the force kernel there).   scripts/pile_sq.sh [particles] [ticks]
export TMPDIR=/tmp
N=${1:-1048576}; T=${2:-450}
OUT=gpurun_out/pile_sq
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python scripts/pile_trace.py $N $T > $OUT/p$i.log 2>&1 || { echo "set $i failed"; tail -3 $OUT/p$i.log; }
done
python - $OUT $N <<'PY'
import collections, csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out}/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, c in sorted(rows.items()):
    v = {name: sum(x for _, x in sorted(vals)[-50:]) / min(50, len(vals)) for name, vals in c.items()}
    if len(c.get("SQ_INSTS_VALU", [])) < 50: continue
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"{k[:60]:60s} cycles {g:8.0f}  VALU wave-insts/particle {v['SQ_INSTS_VALU']/n:6.2f}  VALU busy {4*v['SQ_ACTIVE_INST_VALU']/1024/max(g,1):5.2f}  "
          f"any-inst busy {4*v.get('SQ_ACTIVE_INST_ANY',0)/1024/max(g,1):5.2f}  LDS busy {4*v.get('SQ_ACTIVE_INST_LDS',0)/1024/max(g,1):5.2f}  waves/SIMD {4*v['SQ_WAVE_CYCLES']/1024/max(g,1):5.2f}")
PY
