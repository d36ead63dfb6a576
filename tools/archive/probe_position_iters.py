#!/usr/bin/env python3
"""Scratch probe (CPU): how many of the <= 60 position iterations of b2Island::Solve do the creatures use, and do the
islands that never pass the tolerance test reach an exact fixed point (an iteration that changes no bit of any
position, after which every further iteration is a no-op)?  Oracle compiled with -DREM2D_ORACLE_PROBE into /tmp.
Usage: probe_position_iters.py [n_creatures] [steps] [workload]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
hist = (C.c_int * 160)()
for k in sorted(groups):
    m = Morphology.from_specs(groups[k], k)
    O.lib().rem2d_oracle_probe_pos(hist, 1)
    O.batch_run(ot, m.as_dict(), steps, n_threads=8, flags=O.FLAG_CONTINUOUS)
    O.lib().rem2d_oracle_probe_pos(hist, 1)
    h = np.array(hist[:], dtype=np.int64)
    ok, bad = h[:64], h[64:128]
    tot = ok.sum() + bad.sum()
    print("lanes %2d creatures %5d island-solves %8d: passed the test %.3f (mean iterations %.2f); never passed %.3f" %
          (k, len(groups[k]), tot, ok.sum() / tot, (ok * np.arange(64)).sum() / max(1, ok.sum()), bad.sum() / tot))
    if bad.sum():
        cum = np.cumsum(bad[:63]) / bad.sum()
        print("      of those: exact fixed point by iteration 2/5/10/20/40/59: %s ; none within 60: %.3f ; failing at the end: contacts %.3f joints %.3f both %.3f" %
              ([round(float(cum[i]), 3) for i in (2, 5, 10, 20, 40, 59)], bad[63] / bad.sum(), h[128] / bad.sum(), h[129] / bad.sum(), h[130] / bad.sum()))
