"""The CPU oracle against golden vectors captured from the unmodified reference
(tests/golden/make_golden.py).  This is what pins the oracle: every later parity
claim (HIP vs oracle) rests on these passing."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle.neighbors import neighbor_lists, strip_sort
from oracle.scene import OracleCrate
from oracle.tick import BodyState, closest_points_on_segments, tick_core
from oracle.world import load_scene, pad_segments

RTOL = 1e-5  # BASELINE.json north_star: float positions/velocities within 1e-5 relative


def close(a, b, rtol=1e-9, atol=1e-12):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", golden_names("nbr_"))
def test_strip_sort_and_neighbors_bit_exact(name):
    g = load_golden(name)
    rows, order = strip_sort(g["particles"], float(g["diameter"]))
    assert np.array_equal(rows, g["y_floored"])
    assert np.array_equal(order, g["sorted_indices"])
    counts, table = neighbor_lists(g["particles"], float(g["diameter"]))
    assert np.array_equal(counts, g["counts"])
    assert np.array_equal(table, g["table"])


@pytest.mark.parametrize("name", ["dist_row", "dist_wave"])
def test_point_segment_distance(name):
    g = load_golden(name)
    near, dist = closest_points_on_segments(g["particles"], g["segments"])
    assert np.array_equal(near, g["nearest"])
    assert np.array_equal(dist, g["distances"])


def test_pad_segments():
    g = load_golden("pad_wave")
    assert np.array_equal(pad_segments(g["segments"], float(g["pad"])), g["padded"])


def bodies_of(g):
    return [BodyState(g["body_position"][b], g["body_velocity"][b], float(g["body_omega"][b]), int(g["body_nseg"][b]))
            for b in range(len(g["body_nseg"]))]


def coef_of(g):
    c = {k[5:]: (g[k] if g[k].ndim else float(g[k])) for k in g if k.startswith("coef_")}
    return c


@pytest.mark.parametrize("name", golden_names("tick_"))
def test_single_tick_matches_reference(name):
    g = load_golden(name)
    out = tick_core(g["in_particles"], g["in_velocities"], g["segments"], bodies_of(g), coef_of(g),
                    eta_u01=g["eta_u01"])
    # decisions: bit exact
    assert np.array_equal(out["fixed_positions"], g["fixed_positions"])
    assert np.array_equal(out["wall_count"], g["wall_count"])
    assert np.array_equal(out["neighbor_counts"], g["neighbor_counts"])
    assert np.array_equal(out["neighbor_table"], g["neighbor_table"])
    # floats: far inside the 1e-5 contract (only summation order differs)
    close(out["pressure"], g["out_pressure"])
    close(out["surface_normals"], g["surface_normals"])
    for phase in ("tension", "pressure", "viscosity", "bounce"):
        close(out[f"v_after_{phase}"], g[f"v_after_{phase}"])
    close(out["velocities"], g["out_velocities"])
    close(out["particles"], g["out_particles"])


@pytest.mark.parametrize("scene", ["stirring_cup", "wave_machine"])
def test_scene_trajectory_matches_reference(scene):
    g = load_golden(f"traj_{scene}")
    crate = OracleCrate(load_scene(f"config/{scene}.yaml").world)
    ticks = [int(t) for t in g["ticks"]]
    for t in range(1, max(ticks) + 1):
        crate.physics_tick()
        if t in ticks:
            assert crate.particles.shape == g[f"particles_t{t}"].shape, f"particle count differs at tick {t}"
            close(crate.segments, g[f"segments_t{t}"], rtol=0, atol=0)
            np.testing.assert_allclose(crate.particles, g[f"particles_t{t}"], rtol=RTOL, atol=1e-9)
            np.testing.assert_allclose(crate.particle_velocities, g[f"velocities_t{t}"], rtol=RTOL, atol=1e-7)
            np.testing.assert_allclose(crate.particles_pressure, g[f"pressure_t{t}"], rtol=RTOL, atol=1e-9)


@pytest.mark.parametrize("name", ["tick_synth512_cup", "tick_stirring_cup_t300", "tick_wave_machine_t100"])
def test_loop_structured_tick_matches_reference(name):
    """oracle.tick_loops (what bench.py times as the reference's NumPy path) is the same function."""
    from oracle.tick_loops import tick_loops
    g = load_golden(name)
    eta = g["eta_u01"]
    out = tick_loops(g["in_particles"], g["in_velocities"], g["segments"], bodies_of(g), coef_of(g),
                     eta_source=lambda total: eta[:total])
    close(out["pressure"], g["out_pressure"])
    close(out["velocities"], g["out_velocities"])
    close(out["particles"], g["out_particles"])
