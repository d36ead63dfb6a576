#!/usr/bin/env python3
"""Scratch probe (CPU): at which velocity iteration do the 180 Gauss-Seidel sweeps of b2Island::Solve reach an exact
fixed point (a sweep that changes no bit of any velocity / accumulated impulse)?  Needs the oracle compiled with
-DREM2D_ORACLE_PROBE (done here into /tmp).  Usage: probe_fixed_point.py [n_creatures] [steps] [workload]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/librem2d_oracle_probe.so"
subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                       "-DREM2D_ORACLE_PROBE", "-shared", "-o", so, os.path.join(ROOT, "oracle", "rem2d_oracle.c"), "-lm"])
from oracle import oracle as O  # noqa: E402
O._SO = so
from gym_rem2d_amd import make_terrain, synthetic  # noqa: E402
from gym_rem2d_amd.compiler import Morphology, lanes_for  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
workload = sys.argv[3] if len(sys.argv) > 3 else "lsystem"
hard = workload == "cppn_hardcore"
terrain = make_terrain(4, flat=not hard, hardcore=hard)
xs, ys, polys = terrain.f32()
ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
specs = (synthetic.cppn_specs if hard else synthetic.lsystem_specs)(range(n))
groups = {}
for s in specs:
    groups.setdefault(lanes_for(s.n_bodies), []).append(s)
hist = (C.c_int * 512)()
for k in sorted(groups):
    m = Morphology.from_specs(groups[k], k)
    O.lib().rem2d_oracle_probe_hist(hist, 1)
    for (a, b) in ((0, 60), (60, steps)):
        pass
    O.batch_run(ot, m.as_dict(), steps, n_threads=8, flags=O.FLAG_CONTINUOUS)
    O.lib().rem2d_oracle_probe_hist(hist, 1)
    h = np.array(hist[:], dtype=np.int64)
    tot = h.sum()
    cum = np.cumsum(h) / max(1, tot)
    mean = (h * np.arange(512)).sum() / max(1, tot)
    q = [int(np.searchsorted(cum, p)) for p in (0.1, 0.25, 0.5, 0.75, 0.9, 0.99)]
    print("lanes %2d creatures %5d island-solves %8d: mean sweeps %.1f  p10/25/50/75/90/99 = %s  never(180) %.3f" %
          (k, len(groups[k]), tot, mean, q, h[180:].sum() / max(1, tot)))
