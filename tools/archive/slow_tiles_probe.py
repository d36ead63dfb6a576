#!/usr/bin/env python3
"""Scratch probe (GPU box, -DREM2D_V4_PROBES build via REM2D_LIB_PATH, REM2D_V4_DBG=32): the slowest tiles of the velocity
kernel in one lane bucket of config 3 (argument: lanes, default 8) -- ticks, contact sub-slots, cycles per slot, and what their creatures look like."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

morphs, desc = bench.build_population("lsystem", 65536, 0)
terrain = make_terrain(4, flat=True)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = [q for q in morphs if q.lanes == K][0]
w = BatchedWorld(m.n_envs, m.lanes, flags=1)
w.set_terrain(terrain)
w.reset(m)
w.step(100)
torch.cuda.synchronize()
ev = w.view("toievents").cpu().numpy().reshape(-1).astype(np.float64)
tiles = np.asarray(w.tiles)
t = tiles[:-1]
ok = (np.diff(tiles) >= 4) & (t + 4 <= m.n_envs)
t, te = t[ok], tiles[1:][ok]
tj, tc, ta, packed = (ev[t + i] for i in range(4))   # REM2D_V4_DBG=32: cycles / 64, sub-slots << 16 | ticks
tj, tc, ta = tj * 64, tc * 64, ta * 64
ns, nk = np.floor(packed / 65536), packed % 65536
nt = (w.view("cinfo").cpu().numpy() & 0xff) > 0           # [slots, envs, lanes] touching
man = nt.sum(0)                                            # manifolds per body
jr = m.arrays["jround"].reshape(m.n_envs, K)
par = m.arrays["parent"].reshape(m.n_envs, K)
order = np.argsort(-ta)
print("tiles %d: loop cycles mean %.0f p50 %.0f p99 %.0f max %.0f" % (len(t), ta.mean(), np.percentile(ta, 50), np.percentile(ta, 99), ta.max()))
print("per iteration (180): ticks = period; sub-slots = contact sub-slots executed")
for i in list(order[:8]) + list(order[len(order) // 2: len(order) // 2 + 3]):
    a, b = t[i], te[i]
    P = int(((jr[a:b] >> 16) & 0xff).max())
    per_creature = man[a:b]
    print("  loop %.2f M cycles: period %d, ticks/iter %.2f, sub-slots/iter %.2f, joint slot %.0f cyc, sub-slot %.0f cyc | manifolds per creature %s, max on a body per creature %s, bodies %s"
          % (ta[i] / 1e6, P, nk[i] / 180.0, ns[i] / 180.0, tj[i] / max(nk[i], 1), tc[i] / max(ns[i], 1),
             per_creature.sum(1).tolist(), per_creature.max(1).tolist(), (par[a:b] >= 0).sum(1).tolist()))
