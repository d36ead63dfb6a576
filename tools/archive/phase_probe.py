#!/usr/bin/env python3
"""Scratch probe (GPU box): cost split of the step kernel by ablating iteration counts via step_ex."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

workload = sys.argv[1] if len(sys.argv) > 1 else "lsystem"
morphs, desc = bench.build_population(workload, 65536, 0)
terrain = make_terrain(4, flat=True)
for m in morphs:
    if m.lanes != max(mm.lanes for mm in morphs):
        continue
    w = BatchedWorld(m.n_envs, m.lanes)
    w.set_terrain(terrain)
    w.reset(m)
    w.step(80)
    torch.cuda.synchronize()
    nb = m.n_bodies
    print("bucket lanes=%d envs=%d mean bodies %.1f" % (m.lanes, m.n_envs, nb.mean()))
    jr = (m["jround"] & 0xff).reshape(m.n_envs, m.lanes).max(1) + 1
    per = (m["jround"] >> 16).reshape(m.n_envs, m.lanes).max(1)
    print(" pipeline period per creature: hist", np.bincount(per))
    print(" rounds per creature: mean %.1f max %d hist %s" % (jr.mean(), jr.max(), np.bincount(jr)))
    for (vi, pi) in ((180, 60), (0, 60), (180, 0), (0, 0), (180, 60)):
        torch.cuda.synchronize(); t0 = time.time()
        w.step_ex(10, 1.0 / 50, vi, pi)
        torch.cuda.synchronize(); dt = time.time() - t0
        print("  vel=%3d pos=%2d : %.2f ms/step  positers mean %.1f" % (vi, pi, dt / 10 * 1e3, w.view("positers").float().mean().item()))
    ct = w.view("cinfo").cpu().numpy() & 0xff
    touching = (ct > 0).sum(0)
    print(" touching contacts per body hist:", np.bincount(touching.reshape(-1)))
    print(" pairs per body hist:", np.bincount(w.view("ccount").cpu().numpy().reshape(-1)))
