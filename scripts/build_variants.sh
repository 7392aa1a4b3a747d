#!/bin/bash
# Build variants of libsandcrate_hip.so HERE (hipcc cross-compiles) into scratch/variants/<name>.so so that a
# gpurun call only has to time them:  scripts/build_variants.sh name1:"-DX -DY" name2:""  ...
cd "$(dirname "$0")/.."
mkdir -p scratch/variants
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -Iinclude -Isand_crate_amd/csrc $flags \
      sand_crate_amd/csrc/sandcrate_hip.hip -o scratch/variants/$name.so -ldl 2> scratch/variants/$name.err \
      && echo "built $name" || { echo "BUILD FAILED $name"; tail -5 scratch/variants/$name.err; } ) &
  pids+=($!)
  if (( ${#pids[@]} >= 4 )); then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
