#!/usr/bin/env python3
"""Round 5 probe: where do the wave-instructions of rem2d_velpost_kernel go?  Config 3's population (bench.py's own builders)
is settled normally; then a launch option is switched on for the LAST `--steps` env-steps and the SQ counters of exactly those
launches are read (tools/r05_ablate.sh runs this under `rocprofv3 --pmc` and reduces with collect_profiles.py --tail).

  --debug N        REM2D_OPT debug bits of a -DREM2D_V4_PROBES build: 1 = no contact sub-slots, 2 = no joint slots in vel4
  --pos-iters N    position iterations of the last steps (rem2d_groups_step_ex), 60 = Box2D's; 0 = the loop's share of post

The ablated steps compute WRONG physics (that is the point: only instruction counts are read)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--settle", type=int, default=140)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--debug", type=int, default=0)
    ap.add_argument("--envs", type=int, default=65536)
    a = ap.parse_args()
    import torch
    prep = bench.build_population("lsystem", a.envs, 0)
    morphs, _ = bench.finish_population(prep)
    dev = torch.device("cuda", 0)
    env = bench.make_env(morphs, dev, False, True, False)
    env.step(a.settle)
    torch.cuda.synchronize()
    if a.debug:
        for w, _ in env.worlds:
            w.set_option("debug", a.debug)
    env.step(a.steps)
    torch.cuda.synchronize()
    print("done", a)


if __name__ == "__main__":
    main()
