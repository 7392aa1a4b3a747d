#!/bin/bash
# extra SQ / SQC counters (instruction cache, fetch) for one-off questions; same conventions as collect_sq.sh
export TMPDIR=/tmp
N=${1:-1048576}
OUT=gpurun_out/sqx
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_INSTS_BRANCH SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python bench.py --particles $N --cpu-sample 0 --no-kernel-events --repeats 1 --steps 20 --warmup 5 > $OUT/p$i.log 2>&1 || { echo "set $i failed"; tail -3 $OUT/p$i.log; }
done
python - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void sc::", "")[:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k, {c: round(agg[k][c] / cnt[k][c]) for c in sorted(agg[k])})
PY
