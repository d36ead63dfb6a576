#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld
morphs, desc = bench.build_population("lsystem", 16384, 0)
terrain = make_terrain(4, flat=True)
for m in morphs:
    w = BatchedWorld(m.n_envs, m.lanes); w.set_terrain(terrain); w.reset(m)
    first = {}
    for t in range(0, 200, 10):
        w.step(10)
        px = w.view("px").cpu().numpy(); cc = w.view("ccount").cpu().numpy()
        vy = w.view("vy").cpu().numpy(); vx = w.view("vx").cpu().numpy()
        bad = (~np.isfinite(px)).any(1) | (np.abs(px) > 500).any(1) | (cc >= 9).any(1) | (np.abs(vx) > 90).any(1) | (np.abs(vy) > 90).any(1)
        for e in np.nonzero(bad)[0]:
            first.setdefault(int(e), t + 10)
    print("lanes", m.lanes, "envs", m.n_envs, "suspicious", len(first), "err", np.bincount(w.view("err").cpu().numpy()))
    for e, t in list(first.items())[:6]:
        nb = m.n_bodies[e]
        print("  env", e, "first flagged at step", t, "bodies", nb, "max|v|", float(np.abs(w.view("vx")[e].cpu().numpy()).max()), "px", w.view("px")[e].cpu().numpy()[:nb], "cc", w.view("ccount")[e].cpu().numpy()[:nb])
    np.save("gpurun_out/sus_%d.npy" % m.lanes, np.array(list(first.items())))
