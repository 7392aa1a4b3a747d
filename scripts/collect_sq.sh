#!/bin/bash
# SQ (shader) counters of every kernel of the tick, per launch: instruction counts, VALU / LDS busy time, waits.
# Separate --pmc passes (8 SQ slots per pass; GRBM in its own), --kernel-trace only.
# Run on the MI355X box from the repository root:  scripts/collect_sq.sh [particles]  -> profiles/$SC_PROFILE_TAG_sq_<N>.json (default tag r04)
export TMPDIR=/tmp
N=${1:-1048576}
OUT=gpurun_out/sq
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python bench.py --particles $N --cpu-sample 0 --no-kernel-events --repeats 1 --clock-warmup 0 --steps 20 --warmup 5 > $OUT/p$i.log 2>&1 || { echo "set $i failed"; tail -3 $OUT/p$i.log; }
done
python scripts/summarize_sq.py $OUT $N
