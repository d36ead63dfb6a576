#!/usr/bin/env python3
"""Scratch probe (GPU box): device time of rem2d_vel4_kernel per lane bucket alone (HIP events around the kernel)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

workload = sys.argv[1] if len(sys.argv) > 1 else "lsystem"
morphs, desc = bench.build_population(workload, 65536, 0)
terrain = make_terrain(4, flat=True)
for m in morphs:
    w = BatchedWorld(m.n_envs, m.lanes, flags=1)
    w.set_terrain(terrain)
    w.reset(m)
    w.step(80)
    torch.cuda.synchronize()
    w.enable_timing(True)
    w.kernel_time_ms()
    t0 = time.time()
    w.step(20)
    torch.cuda.synchronize()
    wall = (time.time() - t0) / 20 * 1e3
    ms, n = w.kernel_time_ms()
    sizes = np.diff(w.tiles)
    ct = (w.view("cinfo").cpu().numpy() & 0xff) > 0
    per_tile = np.add.reduceat(ct.sum(0).reshape(-1)[: m.n_envs * m.lanes].reshape(m.n_envs, m.lanes).sum(1), w.tiles[:-1][w.tiles[:-1] < m.n_envs])
    print("lanes %2d envs %6d tiles %5d (creatures/tile %.1f) vel4 %.3f ms  step wall %.3f ms  manifolds/tile mean %.1f max %d  >128: %d" %
          (m.lanes, m.n_envs, len(sizes), sizes.mean(), ms / max(1, n), wall, per_tile.mean(), per_tile.max(), (per_tile > 128).sum()))
    w.close()
