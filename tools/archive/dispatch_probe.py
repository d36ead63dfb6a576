#!/usr/bin/env python3
"""Scratch probe (GPU box, -DREM2D_V4_PROBES build via REM2D_LIB_PATH, REM2D_V4_DBG=16): when does every wavefront of a
rem2d_vel4_kernel launch start and end (s_memrealtime, 100 MHz, chip-wide)?  One step group of config 3 in one merged
launch, as bench.py steps it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain, _lib
from gym_rem2d_amd.world import BatchedWorld

morphs, desc = bench.build_population("lsystem", 65536, 0)
terrain = make_terrain(4, flat=True)
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0 / 3.0
ws = []
for m in sorted(morphs, key=lambda q: -q.lanes):
    n = max(64, int(m.n_envs * frac))
    part = m.take(np.arange(n))
    w = BatchedWorld(part.n_envs, part.lanes, flags=1)
    w.set_terrain(terrain)
    w.reset(part)
    ws.append(w)
arr = (C.c_void_p * len(ws))(*[w.h for w in ws])
st = ws[0]._stream()
_lib.check(_lib.lib().rem2d_worlds_step(arr, len(ws), 100, st))
torch.cuda.synchronize()
starts, ends, lanes = [], [], []
for w in ws:
    ev = w.view("toievents").cpu().numpy().reshape(-1).astype(np.int64)
    t = np.asarray(w.tiles)[:-1]
    ok = (np.diff(np.asarray(w.tiles)) >= 2) & (t + 2 <= w.n_envs)
    t = t[ok]
    starts.append(ev[t]); ends.append(ev[t + 1]); lanes.append(np.full(len(t), w.lanes))
s, e, k = np.concatenate(starts), np.concatenate(ends), np.concatenate(lanes)
t0 = s.min()
s, e = (s - t0) / 100.0, (e - t0) / 100.0   # microseconds
print("wavefronts %d (launch order: widest creatures first)" % len(s))
print("start of the wavefronts after the first one (us): p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(np.percentile(s, [50, 90, 99, 100])))
print("end (us): p50 %.1f  p90 %.1f  p99 %.1f  max %.1f ; duration (us): mean %.1f  p50 %.1f p99 %.1f max %.1f" %
      (tuple(np.percentile(e, [50, 90, 99, 100])) + (np.mean(e - s),) + tuple(np.percentile(e - s, [50, 99, 100]))))
for kk in sorted(set(k.tolist()), reverse=True):
    m = k == kk
    print("  lanes %2d: %5d wavefronts, start p50 %.1f max %.1f us, duration mean %.1f max %.1f us, end max %.1f us" %
          (kk, m.sum(), np.percentile(s[m], 50), s[m].max(), (e - s)[m].mean(), (e - s)[m].max(), e[m].max()))
late = np.argsort(e)[-5:]
print("the five wavefronts that end last: " + ", ".join("lanes %d start %.0f dur %.0f end %.0f" % (k[i], s[i], e[i] - s[i], e[i]) for i in late))
