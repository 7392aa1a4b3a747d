#!/bin/bash
# build the library with different -D switches and time the kernels (60 uniform ticks)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for v in "$@"; do
  flags=$(echo "$v" | tr ',' ' ')
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -Iinclude -Isand_crate_amd/csrc $flags sand_crate_amd/csrc/sandcrate_hip.hip -o sand_crate_amd/libsandcrate_hip.so 2> gpurun_out/variant_build.err || { echo "BUILD FAILED $v"; tail -5 gpurun_out/variant_build.err; continue; }
  python scripts/${TIMER:-quick_time.py} "$v" ${N:-1048576}
done
