import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from gym_rem2d_amd import _lib
prep = bench.build_population("lsystem", 65536, 0)
morphs, _ = bench.finish_population(prep)
env = bench.make_env(morphs, torch.device("cuda", 0), False, True, True)   # discrete: TOI leaves the counters alone
env.step(200)
torch.cuda.synchronize()
for w, _ in env.worlds:
    ev = w.view("toievents").cpu().numpy().reshape(-1)
    ev = ev[(ev >> 30) & 1 == 1]
    if len(ev) == 0: continue
    P = ev & 15; cf = (ev >> 4) & 15; subs = (ev >> 8) & 255; csubs = (ev >> 16) & 255
    print("lanes %2d: %5d tiles, period mean %.2f, conflict phases per period %.2f (of P), sub-slots per iteration %.2f of which in conflict phases %.2f; tiles with NO conflict phase %.1f %%, with all phases in conflict %.1f %%"
          % (w.lanes, len(ev), P.mean(), cf.mean(), subs.mean(), csubs.mean(), 100.0 * (cf == 0).mean(), 100.0 * (cf == P).mean()))
