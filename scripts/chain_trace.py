"""A short run of a 2-slab in-process chain, for a kernel trace of the halo overlap:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python scripts/chain_trace.py [x|y] [overlap 0|1]; then scripts/chain_trace_show.py <dir>"""
import copy, sys
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import bench, sand_crate_amd as sc
from sand_crate_amd.slab import SlabChain
axis = sys.argv[1] if len(sys.argv) > 1 else "y"
overlap = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
per, nslab = 1048576, 2
n = per * nslab
wc, d = bench.world_for(n)
p, v = bench.synthetic_state(n)
chain = SlabChain(copy.deepcopy(wc), p, v, nslab, noise="counter", noise_seed=1, overlap=overlap, axis=axis)
chain.run(12); chain.synchronize()
