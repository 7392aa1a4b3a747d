"""sand_crate_amd: SandCrate's per-timestep particle update on AMD MI355X (gfx950).

Only the hot path of the reference (``Crate.physics_tick`` and its callees) lives here:

    Crate, load_config           the reference's entry points, GPU-backed
    Engine                       one GPU context of libsandcrate_hip.so (C ABI: include/sandcrate_hip.h)
    detect_particle_collisions   the reference's neighbor search, stand-alone
    points_to_segments_distance  the reference's point/segment distance, stand-alone

The HIP library is built in-tree by ``python -m sand_crate_amd.build``.  Nothing here falls back to
the CPU; importing is cheap, the first call that needs the GPU raises if the library is absent.
"""
from .collision_detector import detect_particle_collisions, neighbor_search, strip_sort_particles
from .crate import Crate
from .engine import Engine
from .load_config import Config, PlaybackConfig, WorldConfig, load_config
from .utils.geometry_utils import pad_segments, points_to_segments_distance

__all__ = ["Crate", "Engine", "Config", "PlaybackConfig", "WorldConfig", "load_config", "detect_particle_collisions",
           "neighbor_search", "strip_sort_particles", "pad_segments", "points_to_segments_distance"]
