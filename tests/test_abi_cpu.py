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
