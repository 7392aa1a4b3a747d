"""Headless driver: the reference's CLI (``src/main.py:19-40``) and frame loop
(``src/playback.py:51-65``) without a window.

    python -m sand_crate_amd.main config/wave_machine.yaml [play_recording_dir] [--variants 1] [--ticks N]

Like the reference it walks the 48 coefficient combinations of ``options`` (main.py:10-16, :26-36) --
mutating the loaded config in place, variant after variant -- and runs ``ticks_to_record`` ticks
for each.  Instead of rendering, a variant's recording is the particle state itself: where the
reference writes config.yaml + AVI + GIF (playback.py:109-118) this writes config.yaml + state.npz
(positions, pressure and segments every ``--record-every`` ticks), the state dump the reference
left commented out (playback.py:112-113).  ``--checkpoint-every K`` also writes resumable checkpoints
(``checkpoint_<tick>.npz``: `Crate.begin_checkpoint` captures the state on the device and sends it to pinned host
memory on a side stream while the following ticks run); ``--resume FILE`` continues such a run.
"""
from __future__ import annotations

import argparse
import time
from datetime import datetime
from itertools import product
from pathlib import Path
from typing import Optional

import numpy as np
import yaml

from .crate import Crate
from .load_config import Config, load_config

options = {
    "pressure_amplifier": [20, 40],
    "ignored_pressure": [0.3, 0.1],
    "viscosity": [4, 8],
    "surface_smoothing": [40, 100],
    "target_pressure": [-5, -2, 2],
}


def config_options(options: dict, config: Config):
    """Every combination of the listed coefficient values, written into the SAME config object
    (the reference's generator semantics, main.py:26-36)."""
    names = list(options)
    for values in product(*(options[name] for name in names)):
        for name, value in zip(names, values):
            config.world_config.coefficients[name] = value
        yield config


def deep_dictify(obj):
    """Plain-data view of a config for yaml.safe_dump (objects_utils.py:21-33)."""
    if isinstance(obj, (str, int, float)) or obj is None:
        return obj
    if isinstance(obj, Path):
        return str(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, (list, tuple)):
        return [deep_dictify(x) for x in obj]
    if isinstance(obj, dict):
        return {str(k): deep_dictify(v) for k, v in obj.items()}
    return {str(k): deep_dictify(v) for k, v in vars(obj).items()}


class HeadlessPlayback:
    """`Playback` minus pygame: owns a `Crate`, ticks it, records state instead of frames."""

    def __init__(self, config: Config, recording_dir_path: Optional[Path] = None, *, noise: str = "host",
                 record_every: int = 10, device: int = 0, checkpoint_every: int = 0,
                 resume: Optional[Path] = None) -> None:
        self.config = config
        if recording_dir_path is None:
            stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
            self.recording_dir_path = Path(config.playback_config.recording_output_dir_path) / stamp
        else:
            self.recording_dir_path = Path(recording_dir_path)
        self.crate = (Crate.from_checkpoint(resume, device=device) if resume is not None
                      else Crate(config.world_config, noise=noise, device=device))
        self.checkpoint_every = max(int(checkpoint_every), 0)
        self.checkpoints: list[Path] = []
        self._checkpoint_tick = None
        self.record_every = max(int(record_every), 1)
        self.frames: list[dict] = []
        self.done = False
        self.seconds = 0.0

    def run_live_simulation(self, ticks: Optional[int] = None) -> None:
        n = self.config.playback_config.ticks_to_record if ticks is None else ticks
        t0 = time.perf_counter()
        for _ in range(int(n)):
            self.crate.physics_tick()
            if self.checkpoint_every and self.crate.tick % self.checkpoint_every == 0:
                self._collect_checkpoint()        # the previous one has long arrived
                self.crate.begin_checkpoint()     # returns at once; the transfer overlaps the next ticks
                self._checkpoint_tick = self.crate.tick
            if self.crate.tick % self.record_every == 0:
                self.frames.append({"tick": self.crate.tick, "particles": self.crate.particles.copy(),
                                    "pressure": self.crate.particles_pressure.copy(),
                                    "segments": self.crate.segments.copy()})
            if self.done:
                break
        self._collect_checkpoint()
        self.crate.synchronize()
        self.seconds = time.perf_counter() - t0
        if self.config.playback_config.save_recording:
            self.save_recording(self.recording_dir_path)

    def _collect_checkpoint(self) -> None:
        if self._checkpoint_tick is None:
            return
        self.recording_dir_path.mkdir(exist_ok=True, parents=True)
        path = self.recording_dir_path / f"checkpoint_{self._checkpoint_tick:06d}.npz"
        self.crate.finish_checkpoint(path)
        self.checkpoints.append(path)
        self._checkpoint_tick = None

    def save_recording(self, out_dir: Path) -> None:
        out_dir.mkdir(exist_ok=True, parents=True)
        with open(out_dir / "config.yaml", "w") as f:
            yaml.safe_dump(deep_dictify(self.config), f)
        arrays = {}
        for k, frame in enumerate(self.frames):
            arrays[f"particles_{k}"] = frame["particles"]
            arrays[f"pressure_{k}"] = frame["pressure"]
            arrays[f"segments_{k}"] = frame["segments"]
        arrays["ticks"] = np.array([f["tick"] for f in self.frames], dtype=np.int64)
        np.savez_compressed(out_dir / "state.npz", **arrays)


def main(config_file_path, play_recording: Optional[Path] = None, *, variants: Optional[int] = None,
         ticks: Optional[int] = None, noise: str = "host", record_every: int = 10, checkpoint_every: int = 0,
         resume: Optional[Path] = None) -> list[dict]:
    config = load_config(config_file_path=config_file_path)
    summary = []
    for k, variant in enumerate(config_options(options, config)):
        if variants is not None and k >= variants:
            break
        out = Path(play_recording) / f"variant_{k:02d}" if play_recording is not None else None
        playback = HeadlessPlayback(config=variant, recording_dir_path=out, noise=noise, record_every=record_every,
                                    checkpoint_every=checkpoint_every, resume=resume if k == 0 else None)
        playback.run_live_simulation(ticks)
        summary.append({"variant": k, "ticks": playback.crate.tick, "particles": playback.crate.particle_count,
                        "seconds": playback.seconds,
                        "coefficients": {name: variant.world_config.coefficients[name] for name in options}})
        print(f"variant {k}: {summary[-1]['ticks']} ticks, {summary[-1]['particles']} particles, "
              f"{playback.seconds:.2f} s -> {playback.recording_dir_path}")
    return summary


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("config_file_path", type=Path)
    ap.add_argument("play_recording", type=Path, nargs="?", default=None)
    ap.add_argument("--variants", type=int, default=None, help="stop after this many of the 48 combinations")
    ap.add_argument("--ticks", type=int, default=None, help="override playback.ticks_to_record")
    ap.add_argument("--noise", default="host", choices=["host", "host-sync", "counter", "none"])
    ap.add_argument("--record-every", type=int, default=10)
    ap.add_argument("--checkpoint-every", type=int, default=0, help="write a resumable checkpoint every K ticks (0 = never)")
    ap.add_argument("--resume", type=Path, default=None, help="continue the first variant from this checkpoint file")
    a = ap.parse_args()
    main(a.config_file_path, a.play_recording, variants=a.variants, ticks=a.ticks, noise=a.noise,
         record_every=a.record_every, checkpoint_every=a.checkpoint_every, resume=a.resume)
