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


def pytest_collection_finish(session):
    """A process that uses torch on the GPU as well as libsandcrate_hip.so (the in-process slab chain does: its halo
    buffers are torch tensors) must bring torch's HIP runtime up first; the other order leaves torch without a
    device.  So: when GPU tests are about to run, initialise torch.cuda before any of them loads the library."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # noqa: BLE001 - tests that need torch will say so themselves
            pass


def load_golden(name: str):
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix: str):
    return sorted(p.stem for p in GOLDEN.glob(f"{prefix}*.npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden
