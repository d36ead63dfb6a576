#!/usr/bin/env python3
"""Scratch probe (GPU box): s_memtime split of the position solver's tick loop (contact section / joint section /
verdict) over the wavefronts that iterate (nearly) to the end.  Needs a -DREM2D_POS_STAMPS build via REM2D_LIB_PATH."""
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
L = _lib.lib()
L.rem2d_world_debug_words.argtypes = [C.c_void_p, C.c_void_p]
for m in morphs:
    if m.lanes < 4:
        continue
    part = m.take(np.arange(max(64, m.n_envs // 3)))
    w = BatchedWorld(part.n_envs, part.lanes, flags=1)
    w.set_terrain(terrain)
    w.reset(part)
    w.step(100)
    out = (C.c_int32 * 16)()
    L.rem2d_world_debug_words(w.h, out)
    n_steps = 20
    w.step(n_steps)
    L.rem2d_world_debug_words(w.h, out)
    o = np.array(out[:], dtype=np.float64)
    nw = max(o[9], 1)
    c, j, v, a = (o[2:6] * 64) / nw
    print("lanes %2d: %d long wavefront-steps of %d; per wavefront-step: loop %.0f k cycles (max %.0f k) = contact sections %.0f k + joint sections %.0f k + "
          "verdict %.0f k; ticks %.0f, with a contact section %.0f, with a joint section %.0f -> %.0f cycles per contact section, %.0f per joint section, %.0f per verdict"
          % (m.lanes, nw, n_steps * (part.n_envs * part.lanes // 64), a / 1e3, o[10] * 64 / 1e3, c / 1e3, j / 1e3, v / 1e3, o[6] / nw, o[7] / nw, o[8] / nw,
             c / max(o[7] / nw, 1), j / max(o[8] / nw, 1), v / max(o[6] / nw, 1)))
    w.close()
