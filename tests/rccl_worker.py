"""One process, one GPU: the library's RCCL transport with a one-rank communicator whose left and right
neighbor is the rank itself.  Run by tests/test_gpu_parity.py::test_rccl_transport_moves_halo_buffers_in_stream_order."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    import torch
    torch.cuda.init()
    import sand_crate_amd as sc
    from sand_crate_amd.slab import HALO_COLUMNS, HipSlabBackend
    from slab_worker import synthetic_world
    n, cap = 20000, 4000
    wc, p, v = synthetic_world(n, 0.0, 0.1, margin=0.02)
    d = 2 * wc.coefficients["particle_radius"]
    be = HipSlabBackend(n + 4 * cap, cap, 0, "none", 0)
    cols = np.floor(p[:, 0] / d).astype(np.int64)
    lo, hi = int(cols.min()) + 10, int(cols.max()) - 10
    own = (cols >= lo) & (cols < hi)
    be.set_slab(lo, hi, HALO_COLUMNS, True, True)
    be.load(p[own], v[own], np.flatnonzero(own))
    crate = sc.Crate(wc, noise="none")  # only for the coefficient names and values
    coef = {k: getattr(crate, k) for k in ("dt", "particle_radius", "wall_collision_decay", "pressure_amplifier",
                                           "ignored_pressure", "collider_noise_level", "viscosity", "surface_smoothing",
                                           "target_pressure")}
    be.engine.set_params(gravity=crate.gravity, **coef)
    uid = be.engine.comm_unique_id(be.bundled_rccl())
    be.engine.comm_init(uid, 0, 1, be.bundled_rccl())
    be.recv_left.fill_(-7.0)
    be.recv_right.fill_(-7.0)
    torch.cuda.synchronize()
    be.pack()                      # kernel on the stream ...
    be.exchange_rccl(0, 0)         # ... RCCL group right behind it, no host synchronisation in between
    be.engine.synchronize()
    torch.cuda.synchronize()
    sl, sr = be.send_left.cpu().numpy(), be.send_right.cpu().numpy()
    nl, nr = int(sl[:1].view(np.int32)[0]), int(sr[:1].view(np.int32)[0])
    in_left = own & (cols < lo + HALO_COLUMNS)
    in_right = own & (cols >= hi - HALO_COLUMNS)
    assert nl == int(in_left.sum()) > 50 and nr == int(in_right.sum()) > 50, (nl, nr, in_left.sum(), in_right.sum())
    assert np.array_equal(be.recv_left.cpu().numpy(), sl) and np.array_equal(be.recv_right.cpu().numpy(), sr)
    got = sl.reshape(-1, 5)[1:nl + 1]
    assert sorted(got[:, 4].astype(np.int64)) == sorted(np.flatnonzero(in_left))
    # timing of the exchange alone (host enqueue and stream time), for DESIGN.md
    import time
    for _ in range(5):
        be.exchange_rccl(0, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        be.exchange_rccl(0, 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"self exchange of 2 x {be.send_left.numel() * 8} B: enqueue {1e6 * (t1 - t0) / 50:.1f} us, "
          f"total {1e6 * (t2 - t0) / 50:.1f} us per call")
    be.engine.comm_destroy()
    print("RCCL_SELF_EXCHANGE_OK")


if __name__ == "__main__":
    main()
