#!/usr/bin/env python3
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from gym_rem2d_amd import make_terrain, _lib
from gym_rem2d_amd.world import BatchedWorld
wl = sys.argv[1] if len(sys.argv) > 1 else "chain8"
morphs, desc = bench.build_population(wl, 65536, 0)
terrain = make_terrain(4, flat=True)
for m in morphs:
    w = BatchedWorld(m.n_envs, m.lanes, _lib.FLAG_CONTINUOUS); w.set_terrain(terrain); w.reset(m)
    w.step(80); torch.cuda.synchronize()
    e0 = int(w.view("toievents").sum())
    t0 = time.time(); w.step(50); torch.cuda.synchronize(); dt = time.time() - t0
    e1 = int(w.view("toievents").sum())
    nwaves = m.n_envs * m.lanes // 64
    cc = w.view("ccount").cpu().numpy(); sh = w.view("shape").cpu().numpy()
    print("lanes %d envs %d: %.3f ms/step; TOI events/step %.1f (per wave-step %.3f); pairs/body mean %.2f" % (
        m.lanes, m.n_envs, dt / 50 * 1e3, (e1 - e0) / 50, (e1 - e0) / 50 / nwaves, cc[sh != 0].mean()))
