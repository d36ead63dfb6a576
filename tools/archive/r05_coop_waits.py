#!/usr/bin/env python3
"""Round 5 probe (GPU box, a -DREM2D_COOP=1 -DREM2D_COOP_STATS build through REM2D_LIB_PATH): per tile of the 16-lane bucket, the
shader cycles of the velocity loop and the cycles each role spent waiting at the pipeline's barriers X / Y."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
prep = bench.build_population("lsystem", n, 0)
morphs, _ = bench.finish_population(prep)
env = bench.make_env(morphs, torch.device("cuda", 0), False, True, True)
for w, _ in env.worlds:
    w.set_option("rebalance", 0)
env.step(150)
rows = []
for rep in range(10):
    for w, _ in env.worlds:
        w.view("toievents").zero_()
    env.step(1)
    torch.cuda.synchronize()
    for w, _ in env.worlds:
        if w.lanes != 16:
            continue
        ev = w.view("toievents").cpu().numpy().reshape(-1).astype(np.int64)
        nt = len(ev) // 4
        ev = ev[:nt * 4].reshape(nt, 4)
        head = ev[:, 0]
        ok = (head >> 30) & 1 == 1
        P, cf, subs, csubs = head & 15, (head >> 4) & 15, (head >> 8) & 255, (head >> 16) & 255
        allJ, allC = (ev[:, 1] & 0xffff) * 256.0, (ev[:, 1] >> 16) * 256.0
        xJ, xC = (ev[:, 2] & 0xffff) * 256.0, (ev[:, 2] >> 16) * 256.0
        yJ, yC = (ev[:, 3] & 0xffff) * 256.0, (ev[:, 3] >> 16) * 256.0
        rows.append(np.stack([P, cf, subs, csubs, allJ, allC, xJ, xC, yJ, yC], 1)[ok & (allJ > 0)])
r = np.concatenate(rows)
print("16-lane tiles: %d tile-steps; period %.2f, conflict phases %.2f, sub-slots per iteration %.2f (%.2f in conflict phases)" % (len(r), r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean()))
print("velocity loop: J role %.0f k cycles (waiting at X %.0f k = %.1f %%, at Y %.0f k = %.1f %%); C role %.0f k cycles (waiting at X %.0f k = %.1f %%, at Y %.0f k = %.1f %%)"
      % (r[:, 4].mean() / 1e3, r[:, 6].mean() / 1e3, 100 * r[:, 6].sum() / r[:, 4].sum(), r[:, 8].mean() / 1e3, 100 * r[:, 8].sum() / r[:, 4].sum(),
         r[:, 5].mean() / 1e3, r[:, 7].mean() / 1e3, 100 * r[:, 7].sum() / r[:, 5].sum(), r[:, 9].mean() / 1e3, 100 * r[:, 9].sum() / r[:, 5].sum()))
heavy = r[r[:, 4] >= np.percentile(r[:, 4], 99)]
print("slowest 1 %%: J role %.0f k cycles (X %.1f %%, Y %.1f %%), C role working %.1f %% of its loop; sub-slots per iteration %.2f (%.2f in conflict phases), period %.2f"
      % (heavy[:, 4].mean() / 1e3, 100 * heavy[:, 6].sum() / heavy[:, 4].sum(), 100 * heavy[:, 8].sum() / heavy[:, 4].sum(),
         100 * (1 - (heavy[:, 7].sum() + heavy[:, 9].sum()) / heavy[:, 5].sum()), heavy[:, 2].mean(), heavy[:, 3].mean(), heavy[:, 0].mean()))
