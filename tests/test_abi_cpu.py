"""CPU-side checks of the drop-in boundary: the shared library builds, loads, and exports every
symbol include/sandcrate_hip.h declares; the ctypes table covers the same set.  No compute calls
(there is no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "sandcrate_hip.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from sand_crate_amd import build
    path = build.build()
    return ctypes.CDLL(str(path))


def test_header_declares_the_path():
    names = declared_symbols()
    for must in ("sc_create", "sc_step_begin", "sc_step_finish", "sc_step", "sc_download_state", "sc_neighbor_search",
                 "sc_points_to_segments"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_ctypes_table_matches_header():
    from sand_crate_amd import _native
    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_abi_version_and_error_string(lib):
    lib.sc_abi_version.restype = ctypes.c_int
    assert lib.sc_abi_version() == 5
    lib.sc_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.sc_last_error(), bytes)


def test_struct_layouts_match_header():
    from sand_crate_amd import _native as N
    assert ctypes.sizeof(N.Params) == 11 * 8
    assert ctypes.sizeof(N.Body) == 5 * 8 + 8
    assert ctypes.sizeof(N.Stats) == 2 * 8 + 4 * 4


def test_product_has_no_oracle_dependency():
    """The product path must not route through the CPU oracle."""
    for py in (ROOT / "sand_crate_amd").rglob("*.py"):
        src = py.read_text()
        assert "import oracle" not in src and "from oracle" not in src, py


def test_pad_segments_in_the_library_equals_the_numpy_form():
    """sc_pad_segments (host C inside the library, no GPU involved) against the NumPy restatement of
    geometry_utils.py:146-172 it replaces on the per-tick path: bit for bit, degenerate sizes included."""
    import numpy as np
    from sand_crate_amd.utils.geometry_utils import pad_segments, pad_segments_numpy
    rs = np.random.RandomState(11)
    for n in (0, 1, 3, 8, 16):
        seg = rs.rand(n, 2, 2) * 4.0 - 2.0
        for pad in (0.005, 0.0019086, 0.37, 1e-12):
            a, b = pad_segments(seg, pad), pad_segments_numpy(seg, pad)
            assert a.shape == b.shape == (2 * n, 2, 2) and np.array_equal(a, b)
    # axis-parallel walls (the scenes' edges): exact offsets
    box = np.array([[[0.0, 0.0], [0.0, 1.0]], [[0.0, 0.0], [1.0, 0.0]]])
    assert np.array_equal(pad_segments(box, 0.25), pad_segments_numpy(box, 0.25))


def test_emission_with_a_count_bound_equals_emission_with_the_count():
    """slab.py: draw_with_count_bound -- the draws of crate.py:138-147 taken with an UPPER BOUND of the particle count are
    the draws taken with the count itself as long as no source fills its room; when one does, the stream is rewound and
    the exact count decides.  Either way: same particles, same stream position afterwards."""
    import numpy as np
    from sand_crate_amd.particle_source import build_particle_sources
    from sand_crate_amd.slab import draw_new_particles, draw_with_count_bound
    cfg = [dict(radius=0.3, position=[0.05, 0.95], velocity=[3, 0.0], flow=7000, noise=0.1, active_ticks=500),
           dict(radius=0.1, position=[0.5, 0.5], velocity=[0, 1.0], flow=3000, noise=0.0, active_ticks=50)]
    for max_particles, count, bound in ((4000, 100, 300), (4000, 3990, 3995), (4000, 3999, 4000), (4000, 3970, 4000),
                                        (4000, 4000, 4000), (50, 10, 49)):
        asked = []
        np.random.seed(5)
        want = draw_new_particles(build_particle_sources(cfg), 3, 0.002, max_particles, count)
        after_want = np.random.rand(3)
        np.random.seed(5)

        def exact():
            asked.append(1)
            return count
        got, new_bound = draw_with_count_bound(build_particle_sources(cfg), 3, 0.002, max_particles, bound, exact)
        after_got = np.random.rand(3)
        assert len(got) == len(want) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(got, want))
        assert np.array_equal(after_got, after_want)
        emitted = sum(len(p) for p, _ in want)
        assert new_bound >= count + emitted
        if bound + 40 < max_particles:
            assert not asked  # far from max_particles nobody counts
