#!/usr/bin/env python3
"""GPU probe: what would it buy to keep the creatures of every lane bucket sorted by their CURRENT cost (touching manifolds,
position iterations) instead of by their static schedule key?  A tile / block costs what its most expensive creature costs.
Pass 1: the bench population as benched, `T0` steps; per creature a cost score from the state.  Pass 2: the same population
re-uploaded from reset with every bucket ordered by that score (physics does not depend on the order, so at the same step
numbers the creatures are in the same states), timed over the same steps.   usage: resort_probe.py [T0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402

T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 160
WORKLOAD = os.environ.get("PROBE_WORKLOAD", "lsystem")
morphs, desc = bench.finish_population(bench.build_population(WORKLOAD, 65536, 0))
import torch  # noqa: E402

dev = torch.device("cuda", 0)
hard = WORKLOAD == "cppn_hardcore"


def timed(env, label, pre_steps):
    run = bench.stepper(env, 25)
    run(pre_steps)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        run(20)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) / 20 * 1e3
    print("%-44s %.3f ms/step  %.2f M env-steps/s  err %d" % (label, ms, 65536 / ms / 1e3, int(env.errors().max())), flush=True)


def scores(env, morphs, mode):
    out = [np.zeros(m.n_envs) for m in morphs]
    off = np.cumsum([0] + [m.n_envs for m in morphs])
    for w, idx in env.worlds:
        touch = (w.view("cinfo") & 0xff) > 0
        per_body = touch.sum(dim=0)
        tot = per_body.sum(dim=1).cpu().numpy().astype(np.float64)
        mx = per_body.max(dim=1).values.cpu().numpy().astype(np.float64)
        piters = w.view("positers").cpu().numpy().astype(np.float64)
        period = ((w.view("jround") >> 16) & 0xff).max(dim=1).values.cpu().numpy().astype(np.float64)
        if mode == "pos":
            sc = piters
        elif mode == "vel":
            sc = 7 * period + 10 * np.minimum(2 * mx, tot)
        else:
            sc = 7 * period + 10 * np.minimum(2 * mx, tot) + 0.5 * piters
        pop = idx.cpu().numpy()
        for b in range(len(morphs)):
            sel = (pop >= off[b]) & (pop < off[b + 1])
            out[b][pop[sel] - off[b]] = sc[sel]
    return out


env = bench.make_env(morphs, dev, hard, not hard, False)
timed(env, "static order (as benched), steps %d.." % T0, T0)
sc = {mode: scores(env, morphs, mode) for mode in ("both", "vel", "pos")}
env.close()
for mode in ("both", "vel", "pos"):
    re = [m.take(np.argsort(-s, kind="stable")) for m, s in zip(morphs, sc[mode])]
    env = bench.make_env(re, dev, hard, not hard, False)
    timed(env, "sorted by %s score at step %d" % (mode, T0 + 200), T0)
    env.close()
