"""Shared pytest configuration.

Markers
-------
gpu   needs a real MI355X and the built HIP library; everything else runs on CPU.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix: str):
    return sorted(p.stem for p in GOLDEN.glob(f"{prefix}*.npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden
