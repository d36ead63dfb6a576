#!/usr/bin/env python3
"""Scratch probe (GPU box): steps/s of the single-creature facade `Modular2D.step` (one ABI call, one stream completion
and one host read of the mapped reward / done words per env-step) -- BASELINE config 1's call pattern
(Demo1_Random_Individual.py:4-36) -- next to the oracle on one host thread."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gym_rem2d_amd as G
from gym_rem2d_amd.encodings.direct import DirectEncoding
from gym_rem2d_amd import synthetic

import copy
from gym_rem2d_amd.synthetic import get_module_list
random.seed(0)
env = G.make("Modular2DLocomotion-v0")
env.seed(4)
ml = get_module_list()
tree = copy.deepcopy(DirectEncoding(ml).create(8))
env.reset(tree=tree, module_list=ml)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for _ in range(20):
    env.step(None)
t0 = time.perf_counter()
fit = 0.0
for i in range(n):
    _, r, d, _ = env.step(None)
    if r > 0:
        fit = r
dt = time.perf_counter() - t0
print("Modular2D.step facade: %d steps in %.3f s -> %.0f steps/s (%.1f us per step), fitness so far %.4f" % (n, dt, n / dt, dt / n * 1e6, fit))
