"""YAML scene -> config objects.  Mirrors the reference's ``src/crate/load_config.py:7-46`` so the
two scene files (config/stirring_cup.yaml, config/wave_machine.yaml) load unchanged and a
``WorldConfig`` built by either loader can construct a ``Crate``."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

import yaml


@dataclass
class WorldConfig:
    rigid_bodies: list
    particle_sources: list
    coefficients: dict


@dataclass
class PlaybackConfig:
    save_recording: bool
    ticks_to_record: int
    recording_output_dir_path: Path
    screen_x: int
    screen_y: int


@dataclass
class Config:
    world_config: WorldConfig
    playback_config: PlaybackConfig


def load_config(config_file_path) -> Config:
    raw = yaml.safe_load(Path(config_file_path).read_text())
    world = raw["world"]
    pb = raw["playback"]
    return Config(
        world_config=WorldConfig(
            rigid_bodies=world.get("rigid_bodies", []),
            particle_sources=world.get("particle_sources"),
            coefficients=world.get("coefficients"),
        ),
        playback_config=PlaybackConfig(
            save_recording=pb["save_recording"],
            ticks_to_record=pb["ticks_to_record"],
            recording_output_dir_path=Path(pb["recording_output_dir_path"]),
            screen_x=pb["screen_x"],
            screen_y=pb["screen_y"],
        ),
    )
