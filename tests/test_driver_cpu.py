"""Host logic of the headless driver (no GPU): the coefficient sweep and the config dump."""
from pathlib import Path

import numpy as np
import yaml

ROOT = Path(__file__).resolve().parent.parent


def test_sweep_is_the_references_48_in_place_variants():
    from sand_crate_amd.load_config import load_config
    from sand_crate_amd.main import config_options, options
    cfg = load_config(ROOT / "config" / "wave_machine.yaml")
    seen = []
    for variant in config_options(options, cfg):
        assert variant is cfg  # main.py:26-36 mutates and yields the same object
        seen.append(tuple(variant.world_config.coefficients[k] for k in options))
    assert len(seen) == 48 and len(set(seen)) == 48
    assert seen[0] == (20, 0.3, 4, 40, -5) and seen[-1] == (40, 0.1, 8, 100, 2)
    # untouched coefficients survive the sweep
    assert cfg.world_config.coefficients["dt"] == 0.002


def test_deep_dictify_round_trips_through_yaml():
    from sand_crate_amd.load_config import load_config
    from sand_crate_amd.main import deep_dictify
    cfg = load_config(ROOT / "config" / "stirring_cup.yaml")
    cfg.world_config.coefficients["gravity"] = np.array([0.0, 9.8])
    plain = deep_dictify(cfg)
    back = yaml.safe_load(yaml.safe_dump(plain))
    assert back["world_config"]["coefficients"]["gravity"] == [0.0, 9.8]
    assert back["playback_config"]["ticks_to_record"] == 1200
    assert back["playback_config"]["recording_output_dir_path"] == "../data/recordings"


def test_scene_files_load_unchanged():
    from sand_crate_amd.load_config import load_config
    for name, bodies, cap in (("stirring_cup", 2, 600), ("wave_machine", 2, 4000)):
        cfg = load_config(ROOT / "config" / f"{name}.yaml")
        assert len(cfg.world_config.rigid_bodies) == bodies
        assert cfg.world_config.coefficients["max_particles"] == cap
        assert len(cfg.world_config.particle_sources) == 1


def test_rigid_bodies_match_the_oracle_world():
    """Host-side body placement and motion (rigid_body.py) against the oracle's restatement."""
    from oracle.world import build_bodies, load_scene
    from sand_crate_amd.load_config import load_config
    from sand_crate_amd.rigid_body import build_rigid_bodies
    for name in ("stirring_cup", "wave_machine"):
        mine = build_rigid_bodies(load_config(ROOT / "config" / f"{name}.yaml").world_config.rigid_bodies)
        ref = build_bodies(load_scene(ROOT / "config" / f"{name}.yaml").world.rigid_bodies)
        for _ in range(25):
            for a, b in zip(mine, ref):
                a.apply_velocity(0.002)
                b.advance(0.002)
        for a, b in zip(mine, ref):
            assert np.array_equal(a.segments, b.segments)
            assert np.array_equal(np.asarray(a.center_velocity, float), np.asarray(b.center_velocity, float))
            assert float(a.angular_clockwise_velocity) == float(b.angular_clockwise_velocity)


def test_host_body_motion_and_tick_geometry_match_the_reference():
    """rigid_body.py:42-68 + crate.py:69-71 + geometry_utils.py:146-172 on the host, no GPU: the moving walls
    of both scenes at the golden ticks, bit for bit, and the padded set assembled from per-body halves
    (fixed bodies cached) equal to padding the whole segment array."""
    from sand_crate_amd.crate import tick_geometry
    from sand_crate_amd.load_config import load_config
    from sand_crate_amd.rigid_body import build_rigid_bodies
    from sand_crate_amd.utils.geometry_utils import pad_segments
    for scene in ("stirring_cup", "wave_machine"):
        g = np.load(ROOT / "tests" / "golden" / f"traj_{scene}.npz")
        wc = load_config(ROOT / "config" / f"{scene}.yaml").world_config
        bodies = build_rigid_bodies(wc.rigid_bodies)
        dt, r = wc.coefficients["dt"], wc.coefficients["particle_radius"]
        cache = {}
        for t in range(1, int(g["ticks"].max()) + 1):
            for b in bodies:
                b.apply_velocity(dt)
            seg, pad, packed = tick_geometry(bodies, r, cache)
            if t in g["ticks"]:
                assert np.array_equal(seg, g[f"segments_t{t}"]), (scene, t)
                assert np.array_equal(pad, pad_segments(seg, r))
                assert [n for *_, n in packed] == [len(b) for b in bodies]
