#!/bin/bash
# On the GPU box: time every prebuilt variant (scripts/build_variants.sh) with scripts/quick_time.py.
#   N=1048576 scripts/run_variants.sh name1 name2 ...      (no names: every scratch/variants/*.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
names=("$@")
if (( ${#names[@]} == 0 )); then for f in scratch/variants/*.so; do names+=("$(basename $f .so)"); done; fi
for rep in $(seq 1 ${REPS:-1}); do
  for v in "${names[@]}"; do
    cp scratch/variants/$v.so sand_crate_amd/libsandcrate_hip.so
    python scripts/${TIMER:-quick_time.py} "$v" ${N:-1048576} || echo "FAILED $v"
  done
done
