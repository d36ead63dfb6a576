"""binary32 oracle vs its binary64 "truth" build (oracle/librem2d_oracle_f64.so, same source with every engine
quantity a double) -- SURVEY.md 8c parity protocol (ii)-(iv).

The HIP path equals the binary32 oracle bit for bit (tests/test_parity_gpu.py), so these legs also bound how far the
GPU results are from an exactly rounded engine: trajectories are chaotic once contacts start, therefore the long
horizon is compared as a distribution (fitness statistics, rank correlation), the short horizon point-wise."""
import numpy as np
import pytest

from conftest import oracle_terrain  # noqa: F401


def _terrains(O, terrain):
    xs, ys, polys = terrain.f32()
    p = polys if len(polys) else None
    return O.Terrain(xs, ys, p, terrain.friction), O.Terrain(xs, ys, p, terrain.friction, f64=True)


def _spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)), np.argsort(np.argsort(b))
    return float(np.corrcoef(ra, rb)[0, 1])


def test_f64_build_is_a_different_arithmetic(oracle):
    assert oracle.lib().rem2d_oracle_is_f64() == 0
    assert oracle.lib_f64().rem2d_oracle_is_f64() == 1


def test_short_horizon_free_flight_within_1e4(oracle, flat_terrain):
    """(ii) 100 steps from reset of chains spawned 32 m up: free flight + joints + motors only (no contact).
    binary32 stays within 1e-4 (absolute) of binary64 in x, y, angle."""
    from gym_rem2d_amd import synthetic
    m = synthetic.chain_population(4, 6, "left").as_dict()
    m["y"] = (m["y"] + 30.0).astype(np.float32)
    t32, t64 = _terrains(oracle, flat_terrain)
    a = oracle.batch_run(t32, m, 100, n_threads=2, flags=oracle.FLAG_CONTINUOUS, trace=True)
    b = oracle.batch_run(t64, m, 100, n_threads=2, flags=oracle.FLAG_CONTINUOUS, trace=True)
    assert a["trace"][-1, :, :6, 1].min() > 8.0               # still in the air (lanes 6, 7 are padding)
    d = np.abs(a["trace"].astype(np.float64) - b["trace"].astype(np.float64))
    assert d.max() < 1e-4, d.max()
    assert np.abs(a["bodies"][..., 3:6] - b["bodies"][..., 3:6]).max() < 2e-3   # velocities


def test_long_horizon_fitness_distribution(oracle, flat_terrain):
    """(iii)/(iv) 160 L-system creatures x 600 steps (contacts, TOI, sleeping): individual trajectories diverge
    (chaos), the fitness DISTRIBUTION and the ranking -- what the evolutionary loop consumes -- do not."""
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology, lanes_for
    t32, t64 = _terrains(oracle, flat_terrain)
    groups = {}
    for s in synthetic.lsystem_specs(range(160)):
        groups.setdefault(lanes_for(s.n_bodies), []).append(s)
    f32, f64, dpos = [], [], []
    for k in sorted(groups):
        m = Morphology.from_specs(groups[k], k).as_dict()
        a = oracle.batch_run(t32, m, 600, n_threads=4, flags=oracle.FLAG_CONTINUOUS)
        b = oracle.batch_run(t64, m, 600, n_threads=4, flags=oracle.FLAG_CONTINUOUS)
        f32.append(a["fitness"]); f64.append(b["fitness"])
        dpos.append(np.abs(a["bodies"][:, 0, 0] - b["bodies"][:, 0, 0]))
    f32, f64, dpos = np.concatenate(f32), np.concatenate(f64), np.concatenate(dpos)
    assert _spearman(f32, f64) > 0.97
    assert abs(f32.mean() - f64.mean()) < 0.05 * max(1.0, f64.std())
    for q in (10, 50, 90):
        assert abs(np.percentile(f32, q) - np.percentile(f64, q)) < 0.25
    # most creatures stay close even point-wise; a few diverge visibly (that is the chaos, not an error)
    assert np.median(dpos) < 0.02
