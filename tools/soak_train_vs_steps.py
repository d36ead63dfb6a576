#!/usr/bin/env python3
"""Round 6 soak: the step train against per-step launches, full state `==`, over thousands of launches.

Two BatchedModular2D envs hold the same population -- one steps with the default launch form (REM2D_OPT_FUSE_VELPOST = 2: the step
train, a launch per call with hand-overs between the workgroups of consecutive steps), the other with
REM2D_OPT_FUSE_VELPOST = 1 (one launch per step: the launch boundary is the hand-over) -- and are stepped in lockstep through
calls of random lengths, with the device-made creature order (`rebalance`) on for both.  Every `--compare` launches every field
of every world is compared (`torch.equal` on the arena views); a new population (other seeds) every `--episode` steps.
Modular2DEnv.py:634 is one world.Step: one result, whatever the launch form.

    python tools/soak_train_vs_steps.py [--creatures 8192] [--launches 2400] [--out profiles/r06_train_vs_steps_soak.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--creatures", type=int, default=8192)
    ap.add_argument("--launches", type=int, default=2400)
    ap.add_argument("--max-call", type=int, default=24, help="steps per call: uniform in 1..N")
    ap.add_argument("--episode", type=int, default=2500, help="steps before a new population is uploaded")
    ap.add_argument("--compare", type=int, default=8, help="launches between two full-state comparisons")
    ap.add_argument("--rebalance", type=int, default=50)
    ap.add_argument("--wide", action="store_true")
    ap.add_argument("--tile-shape", type=int, default=-1, help="force a tile shape for both envs (1: the 128-lane flexible shape = "
                                                               "rem2d_step_train128_kernel against per-step launches)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.env import BatchedModular2D
    rng = np.random.default_rng(606)
    launches = steps = compares = mismatches = episodes = 0
    train_launches = 0
    first_bad = None
    t0 = time.time()
    seed0 = 3_000_000
    hardcore = False
    while launches < args.launches:
        seeds = range(seed0, seed0 + args.creatures)
        seed0 += args.creatures
        hardcore = not hardcore
        batches = [(m, idx.tolist()) for m, idx in synthetic.lsystem_batches_native(seeds, n_proc=1)]
        envs = []
        for fuse in (2, 1):
            e = BatchedModular2D(seed=4, hardcore=hardcore, flags=_lib.FLAG_CONTINUOUS, wide=args.wide, options={"fuse_velpost": fuse},
                                 on_handover="raise")
            e.rebalance_every = args.rebalance
            e.step_groups = 1   # (the same worlds on both sides: per-step launches would otherwise be dealt to four step groups)
            if args.tile_shape >= 0:
                e.tile_shape = args.tile_shape
            e._upload(batches, args.creatures)
            envs.append(e)
        # (launch_info: (tile shape, 2 = step train / 1 = velocity + position in one launch per step / 0 = two launches per step))
        assert envs[0].launch_info()[1] == 2 and envs[1].launch_info()[1] in (0, 1), (envs[0].launch_info(), envs[1].launch_info())
        episodes += 1
        done = 0
        while done < args.episode and launches < args.launches:
            n = int(rng.integers(1, args.max_call + 1))
            for e in envs:
                e.step(n)
            done += n
            steps += n
            launches += 1
            train_launches += (n + args.rebalance - 1) // args.rebalance if args.rebalance > 0 else 1
            if launches % args.compare == 0 or done >= args.episode or launches == args.launches:
                torch.cuda.synchronize()
                compares += 1
                assert len(envs[0].worlds) == len(envs[1].worlds)
                for (wa, _), (wb, _) in zip(envs[0].worlds, envs[1].worlds):
                    for name in _lib.FIELDS:
                        a, b = wa.view(name), wb.view(name)
                        if not torch.equal(a, b):
                            # -0.0 == +0.0 and NaN never occurs: compare as numbers before calling it a mismatch
                            if a.dtype.is_floating_point and bool((a == b).all()):
                                continue
                            mismatches += 1
                            if first_bad is None:
                                first_bad = {"field": name, "lanes": wa.lanes, "launch": launches, "episode": episodes,
                                             "differing": int((a != b).sum())}
                assert envs[0].handover_failures() == 0
        for e in envs:
            e.close()
    out = {"tool": "tools/soak_train_vs_steps.py", "creatures": args.creatures, "wide": bool(args.wide), "rebalance": args.rebalance, "tile_shape": args.tile_shape,
           "abi_calls_per_env": launches, "train_kernel_launches": train_launches, "env_steps": steps, "episodes": episodes,
           "full_state_comparisons": compares, "fields_compared": len(_lib.FIELDS), "mismatches": mismatches,
           "first_mismatch": first_bad, "handover_failures": 0, "seconds": round(time.time() - t0, 1),
           "build_id": _lib.build_id(bool(args.wide))}
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(line + "\n")
    return 1 if mismatches else 0


if __name__ == "__main__":
    sys.exit(main())
