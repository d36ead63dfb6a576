#!/usr/bin/env python3
"""Scratch probe (GPU box): wall time of an evaluate() episode of a random L-system population with and without moving
the survivors into smaller worlds (BatchedModular2D.compact), and the step time along the episode."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from gym_rem2d_amd import _lib
from gym_rem2d_amd.env import BatchedModular2D
from gym_rem2d_amd.evaluate import run_episode
from gym_rem2d_amd.population import LSystemPopulation

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
rng = np.random.default_rng(3)
pop = LSystemPopulation.random(n, rng, max_modules=15)
for _ in range(3):
    pop.mutate(0.3, 0.3, 0.2, rng)
batches = pop.compile(0)
flags = _lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN
out = {}
for compact in (False, True, False, True):
    env = BatchedModular2D(flags=flags)
    env._upload(batches, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit = run_episode(env, compact=compact)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = int(env.steps.max())
    print("compact=%s: episode %.3f s, longest creature %d steps, best fitness %.2f, worlds now %s" %
          (compact, dt, steps, float(fit.max()), [w.n_envs for wi, (w, _) in enumerate(env.worlds) if wi not in env._inactive]))
    out[compact] = fit.cpu().numpy()
    env.close()
assert np.array_equal(out[False], out[True])
# step time along the episode (compacting)
env = BatchedModular2D(flags=flags)
env._upload(batches, n)
done = 0
while done < 2500:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(100); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100 * 1e3
    done += 100
    alive = env.compact()
    print("steps %4d: %.3f ms/step, %6d creatures with an open fitness afterwards" % (done, dt, alive))
    if alive == 0:
        break
