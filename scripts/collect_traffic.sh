#!/bin/bash
# HBM-side traffic of every kernel of the tick from rocprofv3 PMC counters, the way
# /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (they do not fit one pass), with --kernel-trace only, unit KiB, and a
# calibration on a known byte count in this code's own access width (8 B per lane).
# Run on the MI355X box from the repository root:  scripts/collect_traffic.sh [particles]
set -e
export TMPDIR=/tmp
N=${1:-262144}
OUT=gpurun_out/traffic
rm -rf $OUT
mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 scripts/traffic_calib.hip -o /tmp/traffic_calib
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -- /tmp/traffic_calib > $OUT/calib_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/bench_$c -- python bench.py --particles $N --cpu-sample 0 --no-kernel-events --repeats 1 --clock-warmup 0 --steps 20 --warmup 5 > $OUT/bench_$c.log 2>&1
done
python scripts/summarize_traffic.py $OUT $N
