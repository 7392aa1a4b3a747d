"""Worker launched by torch.distributed.run: bench.py's N > 1 driver path on the CPU.  `bench.main()` itself runs -- its
argument handling, the slabs it builds (rows, halo overlap requested, the torch.distributed transport), the repetitions
with `reload`, the rank agreement helpers, the JSON line -- with the compute side swapped for the oracle-backed stand-in
backend (tests/slab_oracle_backend.py) and the GPU-only calls of torch.cuda stubbed, under gloo.  The numbers mean
nothing; a Python-level error in that path must not wait for the first 8-GPU run to show.  TEST INFRASTRUCTURE."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    import torch

    import sand_crate_amd as sc
    import sand_crate_amd.slab as slab
    from slab_oracle_backend import OracleSlabBackend

    class Backend(OracleSlabBackend):  # HipSlabBackend's constructor signature
        def __init__(self, capacity, halo_capacity, device, noise, noise_seed):
            super().__init__(halo_capacity, noise, noise_seed)

    class NoCrate:  # the primer / heater contexts of bench.py need a GPU: a do-nothing stand-in
        def __init__(self, *a, **k):
            pass

        def __setattr__(self, name, value):
            object.__setattr__(self, name, value)

        def run(self, n):
            pass

        def synchronize(self):
            pass

    slab.HipSlabBackend = Backend
    sc.Crate = NoCrate
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    import bench
    bench.main()


if __name__ == "__main__":
    main()
