#!/usr/bin/env python3
"""Round 5 probe (GPU box, -DREM2D_V4_PROBES builds through REM2D_LIB_PATH, REM2D_V4_DBG=64): the time every wavefront spends in the
velocity half of rem2d_velpost_kernel (s_memrealtime, 100 MHz) -- single-wavefront form against the cooperative J / C form."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
prep = bench.build_population("lsystem", n, 0)
morphs, _ = bench.finish_population(prep)
env = bench.make_env(morphs, torch.device("cuda", 0), False, True, True)   # discrete: the TOI kernels leave the counters alone
for w, _ in env.worlds:
    w.set_option("rebalance", 0)                                            # (tile t = creatures in arena order: the slots are readable)
env.step(150)
tot = {}
for rep in range(20):
    env.step(1)
    torch.cuda.synchronize()
    for w, _ in env.worlds:
        ev = w.view("toievents").cpu().numpy().reshape(-1).astype(np.int64)
        cap = max(1, min(16, 64 // w.lanes))
        t = ev[::cap][: len(ev) // cap]
        t = t[(t > 0) & (t < 10 ** 7)]
        tot.setdefault(w.lanes, []).append(t)
for lanes in sorted(tot):
    t = np.concatenate(tot[lanes]) / 100.0     # microseconds
    print("lanes %2d: %6d tile-steps, velocity half mean %.1f us, p50 %.1f, p90 %.1f, p99 %.1f, max %.1f" %
          ((lanes, len(t), t.mean()) + tuple(np.percentile(t, [50, 90, 99])) + (t.max(),)))
