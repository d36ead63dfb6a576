#!/usr/bin/env python3
"""Scratch probe (GPU box): env-steps/s of BatchedModular2D.step without per-kernel timing (so that the library replays
its kernel sequence as a hipGraph; REM2D_GRAPH=0 switches that off).  usage: graph_probe.py workload [steps_per_call]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from gym_rem2d_amd import _lib
from gym_rem2d_amd.env import BatchedModular2D

workload = sys.argv[1] if len(sys.argv) > 1 else "lsystem"
spc = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = {"chain4": 4096}.get(workload, 65536)
morphs, _ = bench.build_population(workload, n, 0)
env = BatchedModular2D(flat=True, flags=_lib.FLAG_CONTINUOUS)
batches, lo = [], 0
for m in morphs:
    batches.append((m, list(range(lo, lo + m.n_envs)))); lo += m.n_envs
env._upload(batches, lo)
for _ in range(8):
    env.step(spc)
torch.cuda.synchronize()
steps = 200
t0 = time.perf_counter()
for _ in range(steps // spc):
    env.step(spc)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s graph=%s steps/call=%d: %.2f M env-steps/s, %.3f ms/step, errors %d" % (
    workload, os.environ.get("REM2D_GRAPH", "1"), spc, n * steps / dt / 1e6, dt / steps * 1e3, int(env.errors().max())))
