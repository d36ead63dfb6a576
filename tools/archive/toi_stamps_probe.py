#!/usr/bin/env python3
"""Scratch probe (GPU box): where the TOI solve kernel's latency chain goes.  Needs the diagnostic build
(hipcc ... -DREM2D_TOI_STAMPS -o gym_rem2d_amd/librem2d_stamps.so) selected with REM2D_LIB_PATH."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import _lib, make_terrain
from gym_rem2d_amd.world import BatchedWorld

morphs, _ = bench.build_population("lsystem", 65536, 0)
terrain = make_terrain(4, flat=True)
names = ["alpha pass (b2TimeOfImpact)", "contact updates", "TOI position iterations", "set-up + velocity iterations",
         "integrate / sync fixtures / new pairs"]
for m in morphs:
    w = BatchedWorld(m.n_envs, m.lanes, flags=1)
    w.set_terrain(terrain)
    w.reset(m)
    w.step(80)
    out = (C.c_int32 * 16)()
    L = _lib.lib()
    L.rem2d_world_debug_words.argtypes = [C.c_void_p, C.c_void_p]
    L.rem2d_world_debug_words(w.h, out)
    steps = 20
    w.step(steps)
    L.rem2d_world_debug_words(w.h, out)
    v = np.array(out[:], dtype=np.float64)
    cyc = v[2:7] * 64
    print("lanes %2d: heavy bodies per step %.0f, events per heavy body %.2f (max %d), longest lane %.0f kcycles (max over the %d steps)"
          % (m.lanes, v[11] / steps, v[9] / max(1, v[11]), v[10], v[8] * 64 / 1e3, steps))
    for n, c in zip(names, cyc):
        print("      %-40s %5.1f %%   %.1f kcycles per heavy body" % (n, 100 * c / cyc.sum(), c / max(1, v[11]) / 1e3))
    x = int(v[15])
    print("      longest lane: alpha pass %.0f, velocity %.0f, rest %.0f kcycles; %d sweeps, %d b2TimeOfImpact calls, %d events, island of %d, %d lanes per body"
          % (v[12] * 64 / 1e3, v[13] * 64 / 1e3, v[14] * 64 / 1e3, x & 0xfff, (x >> 12) & 0xff, (x >> 20) & 0xf, (x >> 24) & 7, x >> 27))
    w.close()
