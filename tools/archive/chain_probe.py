#!/usr/bin/env python3
"""Scratch probe (GPU box, -DREM2D_V4_PROBES build via REM2D_LIB_PATH, REM2D_V4_DBG=64, REM2D_TILE_CREATURES such that a
tile is a 64-lane block): per 64-lane block of config 3's lane buckets, the time of its wavefront in the velocity kernel
and in the position kernel of the same step (s_memrealtime, 10 ns) -- what a launch per phase costs (the sum of the two
maxima) against what one launch for both would (the maximum of the sums)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

morphs, desc = bench.build_population("lsystem", 65536, 0)
terrain = make_terrain(4, flat=True)
allv, allp = [], []
for m in morphs:
    K = m.lanes
    if K < 4 or K > 32:
        continue
    w = BatchedWorld(m.n_envs, K, flags=1)
    w.set_terrain(terrain)
    w.reset(m)
    w.step(int(os.environ.get("SETTLE", "100")))
    torch.cuda.synchronize()
    for rep in range(3):
        w.step(1)
        torch.cuda.synchronize()
        ev = w.view("toievents").cpu().numpy().reshape(-1).astype(np.float64)
        tiles = np.asarray(w.tiles)
        per = 64 // K
        nb = m.n_envs // per
        assert np.array_equal(tiles[:nb + 1], np.arange(nb + 1) * per), "tiles are not 64-lane blocks: set REM2D_TILE_CREATURES"
        v = ev[np.arange(nb) * per] * 1e-2      # us
        p = ev[np.arange(nb) * per + 1] * 1e-2
        if rep == 2:
            allv.append(v); allp.append(p)
            print("lanes %2d, %5d blocks: vel4 mean %.0f max %.0f us | post mean %.0f max %.0f us | sum of maxima %.0f, max of sums %.0f us; corr %.2f"
                  % (K, nb, v.mean(), v.max(), p.mean(), p.max(), v.max() + p.max(), (v + p).max(), np.corrcoef(v, p)[0, 1]))
            for name, x in (("vel4", v), ("post", p), ("sum", v + p)):
                o = np.argsort(-x)[:8]
                print("   slowest by %-4s: %s" % (name, ", ".join("#%d %.0f+%.0f" % (i, v[i], p[i]) for i in o)))
            print("   percentiles of vel4 %s | post %s | sum %s" % tuple(np.percentile(x, [50, 90, 99, 99.9]).round().tolist() for x in (v, p, v + p)))
    # over a run of steps: what a barrier per launch costs (sum over steps of the per-step maxima) against what tiles that
    # advance on their own would (the largest per-tile sum), and against the mean tile
    N = int(os.environ.get("RUN", "60"))
    acc_v, acc_p, smax_v, smax_p, smax_vp = np.zeros(nb), np.zeros(nb), 0.0, 0.0, 0.0
    for rep in range(N):
        w.step(1)
        torch.cuda.synchronize()
        ev = w.view("toievents").cpu().numpy().reshape(-1).astype(np.float64)
        v = ev[np.arange(nb) * per] * 1e-2
        p = ev[np.arange(nb) * per + 1] * 1e-2
        acc_v += v; acc_p += p
        smax_v += v.max(); smax_p += p.max(); smax_vp += (v + p).max()
    print("   over %d steps, per step: sum of the launch maxima vel4 %.0f + post %.0f = %.0f us; one launch for both %.0f us; slowest tile on its own %.0f us "
          "(vel4 %.0f + post %.0f); mean tile %.0f us" % (N, smax_v / N, smax_p / N, (smax_v + smax_p) / N, smax_vp / N, (acc_v + acc_p).max() / N,
          acc_v[np.argmax(acc_v + acc_p)] / N, acc_p[np.argmax(acc_v + acc_p)] / N, (acc_v + acc_p).mean() / N))
    o = np.argsort(-(acc_v + acc_p))[:6]
    print("   slowest tiles over the run: " + ", ".join("#%d %.0f+%.0f" % (i, acc_v[i] / N, acc_p[i] / N) for i in o))
    w.close()
v, p = np.concatenate(allv), np.concatenate(allp)
print("all: sum of maxima %.0f us, max of sums %.0f us (vel4 max %.0f, post max %.0f)" % (v.max() + p.max(), (v + p).max(), v.max(), p.max()))
