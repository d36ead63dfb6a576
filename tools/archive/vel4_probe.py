#!/usr/bin/env python3
"""Scratch probe (GPU box): device time of rem2d_vel4_kernel per lane bucket alone (HIP events around the kernel)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

def subslot_stats(w, m, n_sample=200):
    """Host replica of rem2d_vel4_kernel's manifold schedule: sub-slot executions per iteration and tile."""
    K = m.lanes
    tiles = w.tiles
    nt_all = ((w.view("cinfo").cpu().numpy() & 0xff) > 0).sum(0).reshape(-1)[: m.n_envs * K].reshape(m.n_envs, K)
    jr_all = m.arrays["jround"].reshape(m.n_envs, K)
    par_all = m.arrays["parent"].reshape(m.n_envs, K)
    awake = w.view("awake").cpu().numpy().reshape(-1)[: m.n_envs * K].reshape(m.n_envs, K)
    rows = []
    for ti in np.linspace(0, len(tiles) - 2, min(n_sample, len(tiles) - 1)).astype(int):
        a, b = tiles[ti], min(tiles[ti + 1], m.n_envs)
        if b <= a:
            continue
        jr, par, nt = jr_all[a:b], par_all[a:b], nt_all[a:b]
        P = max(1, int(((jr >> 16) & 0xff).max()))
        sub_max = np.zeros(P, dtype=int)
        one_slot = np.zeros(P, dtype=int)
        for c in range(b - a):
            first = np.full(K, 10 ** 6); last = np.full(K, -1)
            for k in range(K):
                if par[c, k] >= 0:
                    r = jr[c, k] & 0xff
                    for x in (k, par[c, k]):
                        first[x] = min(first[x], r); last[x] = max(last[x], r)
            for k in range(K):
                n = nt[c, k]
                if n == 0:
                    continue
                wlo = last[k] if last[k] >= 0 else 0
                wlen = first[k] + P - last[k] if last[k] >= 0 else P
                prev, sub = -1, 0
                for t in range(n):
                    j = (t * wlen) // n
                    sub = sub + 1 if j == prev else 0
                    prev = j
                    ph = (wlo + j) % P
                    sub_max[ph] = max(sub_max[ph], sub + 1)
                oc = (jr[c, k] >> 8) & 0xff
                one_slot[oc % P] = max(one_slot[oc % P], n)
        rows.append((P, sub_max.sum(), one_slot.sum(), nt.max(), (nt > 0).sum(), nt.sum()))
    r = np.array(rows)
    print("   sampled %d tiles: period %.2f, sub-slots/iteration spread %.2f (single slot at offC: %.2f), max manifolds on a body %.2f, "
          "touching bodies %.1f, manifolds %.1f" % (len(r), r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].mean(), r[:, 5].mean()))


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
    if os.environ.get("REM2D_V4_DBG", "0") == "8":
        ev = w.view("toievents").cpu().numpy()
        t = w.tiles[:-1]
        t = t[(np.diff(w.tiles) >= 5) & (t + 5 <= m.n_envs)]
        tj, tc, ta, ns, nk = (ev[t + i].astype(np.float64) for i in range(5))
        t6 = w.tiles[:-1][(np.diff(w.tiles) >= 6) & (w.tiles[:-1] + 6 <= m.n_envs)]
        if len(t6):
            real = ev[t6 + 5].astype(np.float64)            # 100 MHz ticks of the loop
            print("   shader clock inside the loop: %.2f GHz (s_memtime / s_memrealtime)" % ((ev[t6 + 2].astype(np.float64) * 16 / real).mean() * 0.1))
        print("   s_memtime per tile (cycles): joint slots %.0f  contact sub-slots %.0f  loop %.0f | ticks %.0f  sub-slots %.0f -> "
              "%.0f cycles/joint slot, %.0f cycles/sub-slot" % (tj.mean() * 16, tc.mean() * 16, ta.mean() * 16, nk.mean(), ns.mean(),
              (tj * 16 / nk).mean(), (tc * 16 / np.maximum(ns, 1)).mean()))
    subslot_stats(w, m)
    w.close()
