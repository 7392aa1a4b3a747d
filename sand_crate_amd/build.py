"""Builds libsandcrate_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m sand_crate_amd.build
"""
from __future__ import annotations

import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SRC = PKG / "csrc" / "sandcrate_hip.hip"
DEPS = sorted((PKG / "csrc").glob("*")) + [ROOT / "include" / "sandcrate_hip.h"]
LIB = PKG / "libsandcrate_hip.so"

# -ffp-contract=off: float64 decisions must match NumPy's separately rounded multiply/add
# ("hard parts"); the kernels spell out fma() where fusing is harmless.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         f"-I{ROOT / 'include'}", f"-I{PKG / 'csrc'}", "-ldl"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libsandcrate_hip.so cannot be built")
    return exe


def is_stale() -> bool:
    return not LIB.exists() or any(d.stat().st_mtime > LIB.stat().st_mtime for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), *FLAGS, str(SRC), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
