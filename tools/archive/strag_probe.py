#!/usr/bin/env python3
"""GPU probe: what does it buy to give the persistently slow creatures of config 3 a wavefront of their own?

Pass 1: the bench population as benched, 160 steps; per creature a heaviness proxy read back from the state (touching
manifolds in all, the most on one body, schedule period).  Pass 2: the same population with the heaviest creatures moved
into a lane bucket of their own (64 lanes: one creature per 64-lane block in every kernel), then timed like pass 1.
usage: strag_probe.py [fractions ...]   e.g. 0 0.005 0.01 0.02 0.05"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402

fracs = [float(x) for x in sys.argv[1:]] or [0.0, 0.005, 0.01, 0.02, 0.05]
WIDE = int(os.environ.get("STRAG_LANES", "64"))
morphs, desc = bench.finish_population(bench.build_population("lsystem", 65536, 0))
import torch  # noqa: E402
from gym_rem2d_amd.compiler import Morphology  # noqa: E402

dev = torch.device("cuda", 0)


def timed(env, label):
    run = bench.stepper(env, 25)
    run(60)
    run(20)
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        run(20)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) / 20 * 1e3
    print("%-28s %.3f ms/step  %.2f M env-steps/s  worlds %d groups %d err %d" %
          (label, ms, 65536 / ms / 1e3, len(env.worlds), len(env.groups), int(env.errors().max())), flush=True)


env = bench.make_env(morphs, dev, False, True, False)
timed(env, "baseline")
# heaviness after 160 more steps of the same env: per creature (in morphs order)
heavy = []
for m in morphs:
    heavy.append(np.zeros(m.n_envs))
off = np.cumsum([0] + [m.n_envs for m in morphs])
for w, idx in env.worlds:
    touch = (w.view("cinfo") & 0xff) > 0            # [slots, n, K]
    per_body = touch.sum(dim=0)                      # [n, K]
    tot = per_body.sum(dim=1).cpu().numpy().astype(np.float64)
    mx = per_body.max(dim=1).values.cpu().numpy().astype(np.float64)
    period = ((w.view("jround") >> 16) & 0xff).max(dim=1).values.cpu().numpy().astype(np.float64)
    piters = w.view("positers").cpu().numpy().astype(np.float64)
    score = 7 * period + 10 * np.minimum(2 * mx, tot) + 0.25 * piters * (w.lanes >= 8)
    score = score * (w.lanes >= 8)                   # only the 8- and 16-lane creatures make the slow tiles
    pop = idx.cpu().numpy()
    for b in range(len(morphs)):
        sel = (pop >= off[b]) & (pop < off[b + 1])
        heavy[b][pop[sel] - off[b]] = score[sel]
env.close()
allscore = np.concatenate(heavy)
print("score percentiles 50/90/99/99.9/max:", [float(np.percentile(allscore, p)) for p in (50, 90, 99, 99.9, 100)])
for f in fracs:
    if f <= 0:
        continue
    n_move = int(round(f * 65536))
    thr = np.sort(allscore)[-n_move]
    keep, moved = [], []
    for m, h in zip(morphs, heavy):
        sel = h >= thr
        if sel.any() and m.lanes >= 8:
            part = m.take(np.nonzero(sel)[0])
            moved.append(bench._repack(part, WIDE))
            keep.append(m.take(np.nonzero(~sel)[0]))
        else:
            keep.append(m)
    wide = Morphology(sum(p["n_envs"] for p in moved), WIDE)
    for k in wide.arrays:
        wide.arrays[k][:] = np.concatenate([p[k] for p in moved])
    wide.n_bodies[:] = (wide.arrays["shape"].reshape(-1, WIDE) != 0).sum(axis=1)
    env = bench.make_env(keep + [wide], dev, False, True, False)
    timed(env, "top %.2f %% (%d) on %d lanes" % (100 * f, wide.n_envs, WIDE))
    env.close()
