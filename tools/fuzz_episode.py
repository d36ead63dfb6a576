#!/usr/bin/env python3
"""Fuzz on the GPU box, one level up from fuzz_launch_shapes.py: an evaluate() episode of a mixed population through
BatchedModular2D with everything the env may do between two step calls done at random -- step calls of random lengths, compact()
with random thresholds (also while most creatures are still alive and in mid-flight: every state field is moved into new worlds,
rem2d_world_adopt), a random number of step groups, a random cadence of the device-made creature order, hipGraph replay or not.
The float64 fitness of every individual must equal the oracle's.

    python tools/fuzz_episode.py [--rounds 30] [--seed 1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def fuzz(rounds, seed, max_creatures=2500, cap=500, report=print):
    """Returns the number of rounds in which some individual's fitness differed from the oracle's."""
    import torch
    from gym_rem2d_amd import _lib, make_terrain, synthetic
    from gym_rem2d_amd.env import BatchedModular2D
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(seed)
    bad_rounds = 0
    for rnd in range(rounds):
        hard = bool(rnd % 3 == 2)
        terrain = make_terrain(4, hardcore=hard)
        xs, ys, polys = terrain.f32()
        ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
        n = int(rng.integers(200, max_creatures))
        seed0 = int(rng.integers(0, 10 ** 6))
        specs = (synthetic.cppn_specs if hard else synthetic.lsystem_specs)(range(seed0, seed0 + n))
        env = BatchedModular2D(hardcore=hard, flags=_lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN)
        env.step_groups = int(rng.integers(1, 5))
        env.rebalance_every = int(rng.choice([0, 3, 20, 50]))
        env.use_graph = bool(rng.integers(0, 2))
        env.reset_specs(specs)
        T, log = 0, ["groups=%d rebalance=%d graph=%d" % (env.step_groups, env.rebalance_every, env.use_graph)]
        while T < cap:
            k = min(int(rng.choice([1, 5, 25, 60, 100])), cap - T)
            env.step(k)
            T += k
            log.append("step %d" % k)
            if rng.random() < 0.6:
                me, ma = int(rng.choice([1, 64, 2048])), float(rng.choice([0.3, 0.7, 1.0]))
                alive = env.compact(min_envs=me, max_alive=ma)
                log.append("compact(%d, %.1f) -> %d" % (me, ma, alive))
                if alive == 0:
                    break
            elif bool((env.frozen != 0).all()):
                break
        torch.cuda.synchronize()
        fit = env.fitness.cpu().numpy()
        errs = env.errors().cpu().numpy()
        err = int(errs.max())
        uploaded = env._uploaded
        env.close()
        ref = np.full(n, np.nan)
        for m, idx in uploaded:
            ref[np.asarray(idx)] = O.batch_run(ot, m.as_dict(), T, n_threads=os.cpu_count() or 1, flags=O.FLAG_CONTINUOUS)["fitness"]
        # (a creature that overflowed the default build's contact slots is flagged, never silently wrong: only unflagged
        # creatures are held to the oracle here; run_episode re-evaluates the flagged ones in the wide build)
        ok = bool(((fit == ref) | (errs != 0)).all())
        bad_rounds += 0 if ok else 1
        report(json.dumps(dict(round=rnd, workload="network/hardcore" if hard else "lsystem/rough", individuals=n, steps=T, err=err,
                               differing=int((fit != ref).sum()), ok=bool(ok), ops=log if not ok else len(log))))
    return bad_rounds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-creatures", type=int, default=2500)
    ap.add_argument("--cap", type=int, default=500)
    args = ap.parse_args()
    bad = fuzz(args.rounds, args.seed, args.max_creatures, args.cap, report=lambda line: print(line, flush=True))
    print("FUZZ " + ("OK" if bad == 0 else "MISMATCH in %d rounds" % bad))
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
