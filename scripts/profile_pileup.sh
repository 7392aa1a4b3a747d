#!/bin/bash
# The pile-up regime of the contract workload on the MI355X box: kernel trace of a 500-tick run, every kernel
# averaged over its last 50 launches.   scripts/profile_pileup.sh r02 [particles] [ticks]
export TMPDIR=/tmp
TAG=${1:-r02}; N=${2:-1048576}; T=${3:-500}
mkdir -p gpurun_out; rm -rf gpurun_out/${TAG}_pile_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_pile_trace -- python scripts/pile_trace.py $N $T > gpurun_out/${TAG}_pile_trace.log 2>&1
{ echo "rocprofv3 --kernel-trace -- python scripts/pile_trace.py $N $T ; average duration of each kernel over its last 50 launches"; grep "^ticks" gpurun_out/${TAG}_pile_trace.log; python scripts/pile_trace_summary.py gpurun_out/${TAG}_pile_trace 50; } > gpurun_out/${TAG}_pileup_kernels_$N.txt
rm -rf gpurun_out/${TAG}_pile_trace
cat gpurun_out/${TAG}_pileup_kernels_$N.txt
