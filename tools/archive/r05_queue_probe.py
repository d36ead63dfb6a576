import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
prep = bench.build_population("lsystem", 65536, 0)
morphs, desc = bench.finish_population(prep)
import torch
dev = torch.device("cuda", 0)
def run(tag, env_over):
    for k in ("REM2D_FUSE_VELPOST", "REM2D_PROBE_QUEUE", "REM2D_STEP_GROUPS"):
        os.environ.pop(k, None)
    os.environ.update(env_over)
    env = bench.make_env(morphs, dev, False, True, True)
    env.step(60)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); env.step(20); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    fit = env.fitness.cpu().numpy().copy()
    print(tag, "ms per step %.3f %.3f %.3f" % tuple(t / 20 * 1e3 for t in ts), "err", int(env.errors().max()), flush=True)
    env.close()
    return fit
f0 = run("base 4 groups", {})
f1 = run("queue agent-release 1 group", {"REM2D_FUSE_VELPOST": "2", "REM2D_PROBE_QUEUE": "1", "REM2D_STEP_GROUPS": "1"})
print("equal", bool(np.array_equal(f0, f1)))
f2 = run("queue L2-only release 1 group", {"REM2D_FUSE_VELPOST": "2", "REM2D_PROBE_QUEUE": "2", "REM2D_STEP_GROUPS": "1"})
print("equal", bool(np.array_equal(f0, f2)))
f3 = run("queue L2-only release 2 groups", {"REM2D_FUSE_VELPOST": "2", "REM2D_PROBE_QUEUE": "2", "REM2D_STEP_GROUPS": "2"})
print("equal", bool(np.array_equal(f0, f3)))
