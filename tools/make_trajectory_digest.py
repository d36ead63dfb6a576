#!/usr/bin/env python3
"""Writes tests/golden/trajectory_digest.json: SHA-256 digests of the oracle's state (poses, velocities,
sleep timers, reward, done, fitness) after a fixed number of Modular2D.step calls, for seeded synthetic
populations.  The digests pin the oracle across rounds (tests/test_oracle_kat.py) and give the GPU parity
tests a committed golden vector that does not need the oracle at run time (tests/test_parity_gpu.py).

    python tools/make_trajectory_digest.py          # rewrites the fixture from oracle/rem2d_oracle.c
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

CASES = [  # name, population, terrain, flags (1 = continuous physics), steps
    ("chain4_left/flat/discrete", "chain4_left", "flat", 0, 300),
    ("chain8_top/rough/continuous", "chain8_top", "rough", 1, 300),
    ("lsystem_k16/rough/discrete", "lsystem_k16", "rough", 0, 300),
    ("lsystem_k32/rough/continuous", "lsystem_k32", "rough", 1, 300),
    ("direct/flat/continuous", "direct", "flat", 1, 300),
    ("cppn/hardcore/continuous", "cppn", "hardcore", 1, 400),
]


def population(name):
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology
    if name == "chain4_left":
        return synthetic.chain_population(16, 4, "left")
    if name == "chain8_top":
        return synthetic.chain_population(9, 8, "top")
    if name in ("lsystem_k16", "lsystem_k32"):
        ls = synthetic.lsystem_specs(range(48), mutate_odd=True)
        if name == "lsystem_k32":
            return Morphology.from_specs(ls, 32)
        return Morphology.from_specs([s for s in ls if s.n_bodies <= 16], 16)
    if name == "direct":
        return Morphology.from_specs(synthetic.direct_specs(range(40)))
    if name == "cppn":
        return Morphology.from_specs(synthetic.cppn_specs(range(24)))
    raise KeyError(name)


def terrain(name):
    from gym_rem2d_amd import make_terrain
    if name == "flat":
        return make_terrain(4, flat=True)
    if name == "hardcore":
        return make_terrain(4, hardcore=True)
    return make_terrain(4)


def digest(bodies, n_bodies, reward, done, fitness):
    """bodies [N,K,8] f32 (x,y,a,vx,vy,w,sleepT,awake); only the first n_bodies[i] lanes of creature i count."""
    h = hashlib.sha256()
    b = np.ascontiguousarray(bodies, dtype=np.float32) + np.float32(0.0)   # -0.0 -> +0.0
    for i in range(b.shape[0]):
        h.update(b[i, : int(n_bodies[i])].tobytes())
    h.update((np.asarray(reward).astype(np.float32) + np.float32(0.0)).tobytes())   # the C ABI exposes reward as f32
    h.update(np.asarray(done, dtype=np.int32).tobytes())
    h.update((np.asarray(fitness, dtype=np.float64) + 0.0).tobytes())
    return h.hexdigest()


def main():
    from oracle import oracle as O
    O.build()
    out = {"_comment": "SHA-256 of the oracle state after `steps` Modular2D.step calls; see tools/make_trajectory_digest.py",
           "cases": {}}
    for name, pop, ter, flags, steps in CASES:
        m, t = population(pop), terrain(ter)
        xs, ys, polys = t.f32()
        ot = O.Terrain(xs, ys, polys if len(polys) else None, t.friction)
        r = O.batch_run(ot, m.as_dict(), steps, n_threads=os.cpu_count() or 1, flags=flags)
        out["cases"][name] = {"population": pop, "terrain": ter, "flags": flags, "steps": steps,
                              "n_envs": int(m.n_envs), "lanes": int(m.lanes),
                              "root_x_sum": float(np.float64(r["bodies"][:, 0, 0].astype(np.float64).sum())),
                              "sha256": digest(r["bodies"], m.n_bodies, r["reward"], r["done"], r["fitness"])}
        print(name, out["cases"][name]["sha256"][:16], out["cases"][name]["root_x_sum"])
    with open(os.path.join(ROOT, "tests", "golden", "trajectory_digest.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
