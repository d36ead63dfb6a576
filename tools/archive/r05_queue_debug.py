"""Round 5 debug run of the ordered-queue probe (tools/archive/r05_step_tile.patch): one small and one config-3-sized discrete world."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["REM2D_FUSE_VELPOST"] = "2"
os.environ["REM2D_PROBE_QUEUE"] = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["REM2D_PROBE_DEBUG"] = "1"
os.environ["REM2D_STEP_GROUPS"] = "1"
import torch
from gym_rem2d_amd import _lib, make_terrain, synthetic
from gym_rem2d_amd.compiler import Morphology
from gym_rem2d_amd.world import BatchedWorld
specs = [s for s in synthetic.lsystem_specs(range(600), mutate_odd=True) if s.n_bodies <= 8]
morph = Morphology.from_specs(specs, 8)
print('morph ready', morph.n_envs, flush=True)
w = BatchedWorld(morph.n_envs, morph.lanes, 0, "cuda:0")
print('world created', flush=True)
w.set_terrain(make_terrain(4, flat=True))
w.reset(morph)
torch.cuda.synchronize(); print('reset done', flush=True)
for n in (1, 1, 3):
    t0 = time.perf_counter(); w.step(n); torch.cuda.synchronize(); print("small world step(%d): %.3f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
w.close()
if len(sys.argv) > 2:
    import bench
    prep = bench.build_population("lsystem", 65536, 0)
    morphs, desc = bench.finish_population(prep)
    env = bench.make_env(morphs, torch.device("cuda", 0), False, True, True)
    for n in (1, 5, 20, 20):
        t0 = time.perf_counter(); env.step(n); torch.cuda.synchronize(); print("config 3 step(%d): %.3f ms per step" % (n, (time.perf_counter() - t0) * 1e3 / n), flush=True)
    env.close()
