#!/usr/bin/env python3
"""Scratch (GPU box): which creatures of an array population differ from the oracle, in which bucket / tile."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.env import BatchedModular2D
from gym_rem2d_amd.population import LSystemPopulation
from oracle import oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(11)
pop = LSystemPopulation.random(600, rng, max_modules=15)
terrain = make_terrain(4)
xs, ys, polys = terrain.f32()
ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
env = BatchedModular2D()
for name, p in (("orig", pop), ("perm", pop.select(rng.permutation(600)))):
    batches = p.compile(2)
    env.trees = env.robots = None
    env._upload(batches, len(p))
    env.step(steps)
    torch.cuda.synchronize()
    print(name, "errors max", int(env.errors().max()))
    for (w, idx), (m, _) in zip(env.worlds, batches):
        got = w.bodies()
        ref = O.batch_run(ot, m.as_dict(), steps, n_threads=8, flags=1)["bodies"]
        bad = np.where((got != ref).any(axis=(1, 2)))[0]
        tiles = w.tiles
        print("  lanes", w.lanes, "envs", w.n_envs, "tiles", len(tiles) - 1, "bad creatures", len(bad), bad[:20])
        if len(bad):
            t = np.searchsorted(tiles, bad, side="right") - 1
            print("    bad tiles", np.unique(t)[:20], "err of bad", w.view("err").cpu().numpy()[bad][:20])
            for tt in np.unique(t)[:3]:
                a, b = tiles[tt], tiles[tt + 1]
                jr = m.arrays["jround"].reshape(m.n_envs, m.lanes)[a:min(b, m.n_envs)]
                par = m.arrays["parent"].reshape(m.n_envs, m.lanes)[a:min(b, m.n_envs)]
                P = max(1, ((jr >> 16) & 0xff).max())
                print("    tile", tt, "creatures", a, b, "P", P, "periods", np.unique((jr >> 16) & 0xff),
                      "joints/phase", [int(((par >= 0) & ((jr & 0xff) % P == s)).sum()) for s in range(P)],
                      "bad in tile", bad[(bad >= a) & (bad < b)])
                cc = (w.view("cinfo").cpu().numpy() & 0xff)[:, a:min(b, m.n_envs)]
                print("      touching manifolds in tile now", int((cc > 0).sum()))
