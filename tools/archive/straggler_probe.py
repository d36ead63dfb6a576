#!/usr/bin/env python3
"""Scratch probe (GPU box, -DREM2D_V4_PROBES build via REM2D_LIB_PATH, REM2D_V4_DBG=8): per tile, the cycles of the
velocity loop (s_memtime) next to the position iterations its creatures used -- are the slow wavefronts of the velocity
kernel and of the position kernel the same ones?  (What a vel+post kernel without the barrier in between could gain.)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

morphs, desc = bench.build_population("lsystem", 65536, 0)
terrain = make_terrain(4, flat=True)
for m in morphs:
    w = BatchedWorld(m.n_envs, m.lanes, flags=1)
    w.set_terrain(terrain)
    w.reset(m)
    w.step(100)
    torch.cuda.synchronize()
    pos = w.view("positers").cpu().numpy().reshape(-1)[: m.n_envs].astype(np.int64)
    ev = w.view("toievents").cpu().numpy().reshape(-1)
    tiles = np.asarray(w.tiles)
    t = tiles[:-1]
    ok = (np.diff(tiles) >= 5) & (t + 5 <= m.n_envs)
    t = t[ok]
    loop = ev[t + 2].astype(np.float64) * 16          # cycles of the 180-iteration loop of the tile
    ends = tiles[1:][ok]
    pmax = np.array([pos[a:b].max() for a, b in zip(t, ends)])
    pmean = np.array([pos[a:b].mean() for a, b in zip(t, ends)])
    q = lambda x: np.percentile(x, [50, 90, 99, 100]).round(0)
    print("lanes %2d  creatures %6d  tiles %5d" % (m.lanes, m.n_envs, len(t)))
    print("   position iterations per creature: mean %.1f  p50/p90/p99/max %s ; share at 60: %.3f" % (pos.mean(), q(pos), (pos >= 60).mean()))
    print("   per tile: max position iterations mean %.1f  p50/p90/p99/max %s ; tiles with a 60: %.3f" % (pmax.mean(), q(pmax), (pmax >= 60).mean()))
    print("   velocity loop cycles per tile: mean %.0f  p50/p90/p99/max %s" % (loop.mean(), q(loop)))
    print("   correlation(loop cycles, max position iterations) = %.2f" % np.corrcoef(loop, pmax)[0, 1])
    top = loop >= np.percentile(loop, 95)
    print("   the slowest 5 %% of velocity tiles: max position iterations mean %.1f (all tiles %.1f)" % (pmax[top].mean(), pmax.mean()))
    w.close()
