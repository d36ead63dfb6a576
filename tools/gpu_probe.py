#!/usr/bin/env python3
"""Scratch probe (GPU box): HIP path vs oracle on a few populations, prints first mismatch."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from gym_rem2d_amd import synthetic, make_terrain
from gym_rem2d_amd.compiler import Morphology
from gym_rem2d_amd.world import BatchedWorld
from oracle import oracle as O


def compare(name, morph, terrain, steps, chunk=1):
    xs, ys, polys = terrain.f32()
    ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
    w = BatchedWorld(morph.n_envs, morph.lanes)
    w.set_terrain(terrain)
    w.reset(morph)
    md = morph.as_dict()
    ref = O.batch_run(ot, md, steps, n_threads=8, trace=True)
    bad = None
    t0 = time.time()
    for s in range(0, steps, chunk):
        w.step(chunk)
        b = w.bodies()
        tr = ref["trace"][s + chunk - 1]
        if not np.array_equal(b[..., :3], tr):
            d = np.argwhere(b[..., :3] != tr)
            bad = (s + chunk, d[0], b[tuple(d[0][:2])], tr[tuple(d[0][:2])], len(d))
            break
    torch.cuda.synchronize()
    dt = time.time() - t0
    b = w.bodies()
    final_ok = np.array_equal(b, ref["bodies"])
    err = w.view("err").cpu().numpy()
    print("%-28s envs=%d lanes=%d steps=%d  traj %s  final %s  err=%s  reward_ok=%s fit_ok=%s  (%.2fs)" % (
        name, morph.n_envs, morph.lanes, steps, "OK" if bad is None else "MISMATCH@%r" % (bad,), final_ok,
        np.unique(err), np.array_equal(w.view("reward").cpu().numpy().astype(np.float64), ref["reward"].astype(np.float32).astype(np.float64)),
        np.array_equal(w.view("fitness").cpu().numpy(), ref["fitness"]), dt))
    return bad is None


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    flat = make_terrain(4, flat=True)
    rough = make_terrain(4)
    ok = True
    ok &= compare("chain4 top flat", synthetic.chain_population(16, 4, "top"), flat, steps)
    ok &= compare("chain4 left flat", synthetic.chain_population(16, 4, "left"), flat, steps)
    ok &= compare("chain4 right rough", synthetic.chain_population(16, 4, "right"), rough, steps)
    specs = synthetic.lsystem_specs(range(64))
    ok &= compare("lsystem64 flat", Morphology.from_specs(specs, 32), flat, steps)
    ok &= compare("lsystem64 rough", Morphology.from_specs(specs, 32), rough, steps)
    specs = synthetic.direct_specs(range(64))
    ok &= compare("direct64 rough", Morphology.from_specs(specs), rough, steps)
    print("ALL OK" if ok else "FAILURES")
