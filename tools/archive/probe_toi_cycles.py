#!/usr/bin/env python3
"""Scratch probe (CPU): do the 180 velocity sweeps of a TOI sub-step (b2Island::SolveTOI, one body against the ground,
no warm start) run into a cycle -- a state identical to the one p sweeps earlier -- and at which sweep?  From there
on the remaining sweeps are determined by (180 - sweep) mod p.  Oracle compiled with -DREM2D_ORACLE_PROBE into /tmp.
Usage: probe_toi_cycles.py [n_creatures] [steps] [workload]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
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
P = 64
hist = (C.c_int * ((P + 1) * 256))()
tot = np.zeros((P + 1, 256), dtype=np.int64)
for k in sorted(groups):
    m = Morphology.from_specs(groups[k], k)
    O.lib().rem2d_oracle_probe_toi(hist, 1)
    O.batch_run(ot, m.as_dict(), steps, n_threads=8, flags=O.FLAG_CONTINUOUS)
    O.lib().rem2d_oracle_probe_toi(hist, 1)
    tot += np.array(hist[:], dtype=np.int64).reshape(P + 1, 256)
events = tot.sum()
none = tot[0, 255]
print("TOI sub-steps %d: no cycle of period <= %d within 180 sweeps: %.3f" % (events, P, none / events))
by_p = tot[1:].sum(1)
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 64)):
    sel = tot[lo:hi + 1]
    cnt = sel.sum()
    if cnt:
        sweeps = (sel * np.arange(256)[None, :]).sum() / cnt
        print("  period %2d..%2d: %.3f of the sub-steps, first seen at sweep %.1f on average" % (lo, hi, cnt / events, sweeps))
# sweeps needed if the solve stops at the first repeat and runs (180 - sweep) mod p more
need = 0.0
for p in range(1, P + 1):
    for i in range(256):
        if tot[p, i]:
            need += tot[p, i] * (min(i, 179) + 1 + ((179 - min(i, 179)) % p))
need += none * 180
print("mean sweeps per sub-step with exact cycle exits up to period %d: %.1f (period <= 4 only: see DESIGN.md; no exit: 180)" % (P, need / events))
