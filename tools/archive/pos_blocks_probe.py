#!/usr/bin/env python3
"""Scratch probe (GPU box, -DREM2D_POS_STAMPS build via REM2D_LIB_PATH): the slowest 64-lane blocks of the position kernel in
one lane bucket of config 3 (argument: lanes, default 16; discrete stepping so that the TOI kernel leaves the counters alone)
-- ticks, contact sections, cycles per section, and what their creatures look like."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from gym_rem2d_amd import make_terrain
from gym_rem2d_amd.world import BatchedWorld

morphs, desc = bench.build_population("lsystem", 65536, 0)
terrain = make_terrain(4, flat=True)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = [q for q in morphs if q.lanes == K][0]
w = BatchedWorld(m.n_envs, m.lanes, flags=0)
w.set_terrain(terrain)
w.reset(m)
w.step(100)
torch.cuda.synchronize()
ev = w.view("toievents").cpu().numpy().reshape(-1).astype(np.int64)
per = 64 // K
nb = m.n_envs // per
b = np.arange(nb) * per
cyc, packed, cC, cJ = ev[b] * 64.0, ev[b + 1], ev[b + 2] * 64.0, ev[b + 3] * 64.0
ticks, csec = packed & 0xffff, packed >> 16
pit = w.view("positers").cpu().numpy().reshape(-1)
nt = (w.view("cinfo").cpu().numpy() & 0xff) > 0
man = nt.sum(0)
jr = m.arrays["jround"].reshape(m.n_envs, K)
par = m.arrays["parent"].reshape(m.n_envs, K)
print("lanes %d, %d blocks: loop cycles mean %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % ((K, nb, cyc.mean()) + tuple(np.percentile(cyc, [50, 90, 99, 99.9])) + (cyc.max(),)))
order = np.argsort(-cyc)
for i in list(order[:12]) + list(order[nb // 100: nb // 100 + 3]) + list(order[nb // 10: nb // 10 + 2]):
    a = i * per
    P = ((jr[a:a + per] >> 16) & 0xff).max(1)
    R = (jr[a:a + per] & 0xff).max(1)
    print("  block %5d: loop %.0f k cycles, ticks %d, contact sections %d (%.0f cycles each), joint sections %.0f cycles each | periods %s rounds %s "
          "position iterations %s bodies %s manifolds per creature %s, most on a body %s"
          % (i, cyc[i] / 1e3, ticks[i], csec[i], cC[i] / max(csec[i], 1), cJ[i] / max(ticks[i], 1), P.tolist(), R.tolist(), pit[a:a + per].tolist(),
             (par[a:a + per] >= 0).sum(1).tolist(), man[a:a + per].sum(1).tolist(), man[a:a + per].max(1).tolist()))
