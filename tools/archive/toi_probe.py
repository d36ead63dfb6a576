#!/usr/bin/env python3
"""Scratch probe (GPU box): how much TOI work a settled population does per step -- bodies that reach the heavy
kernel (work-list length is not exposed, so: bodies whose pair flags carry a computed TOI) and TOI events."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

morphs, desc = bench.build_population(sys.argv[1] if len(sys.argv) > 1 else "lsystem", 65536, 0)
terrain = make_terrain(4, flat=True)
m = max(morphs, key=lambda mm: mm.lanes)
w = BatchedWorld(m.n_envs, m.lanes, flags=1)
w.set_terrain(terrain)
w.reset(m)
w.step(100)
ev0 = int(w.view("toievents").sum())
rows = []
for _ in range(20):
    w.step(1)
    torch.cuda.synchronize()
    info = w.view("cinfo").cpu().numpy()
    toiflag = ((info >> 17) & 1).sum(0) > 0          # CI_TOIFLAG on any pair of the body
    rows.append((int(toiflag.sum()), int(w.view("toievents").sum())))
ev = [b - a for (_, a), (_, b) in zip([(0, ev0)] + rows[:-1], rows)]
nb = int(m.n_bodies.sum())
print("bodies %d; per step: bodies with a computed TOI mean %.0f (%.2f%%), TOI events mean %.1f" %
      (nb, np.mean([r[0] for r in rows]), 100 * np.mean([r[0] for r in rows]) / nb, np.mean(ev)))
