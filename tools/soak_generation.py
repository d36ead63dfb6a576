#!/usr/bin/env python3
"""One-off whole-episode parity sweep on the GPU box: a generation of random L-system individuals through the
evaluation path an EA uses (native compiler, lane buckets, the tile-shape policy of the population size, creature
order re-made on the device, compaction of the survivors into smaller worlds, wide-slot re-evaluation of overflowing
creatures) against the oracle's float64 fitness of EVERY individual (tests/test_parity_gpu.py checks a 1 % sample).

    python tools/soak_generation.py [--n 131072] [--cap 1000] [--check 1.0]
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
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--cap", type=int, default=1000, help="episode length limit (evaluate()'s range(max_steps))")
    ap.add_argument("--check", type=float, default=1.0, help="fraction of the individuals the oracle re-evaluates")
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--max-modules", type=int, default=15)
    ap.add_argument("--chunk", type=int, default=8192, help="individuals per oracle call (bounds host memory)")
    ap.add_argument("--encoding", choices=["lsystem", "network", "direct", "network_arrays"], default="lsystem",
                    help="network: config 4's generator (one network-encoded creature per seed) on the hardcore track; direct / "
                         "network_arrays: round 5's array populations (DirectPopulation / NetworkPopulation, `--mutations` rounds "
                         "of mutate(0.2, 0.2, 0.2) after construction) on the default terrain")
    ap.add_argument("--mutations", type=int, default=3)
    args = ap.parse_args()
    from gym_rem2d_amd import _lib, make_terrain
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import run_episode
    from gym_rem2d_amd.population import DirectPopulation, LSystemPopulation, NetworkPopulation
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(args.seed)
    hard = args.encoding == "network"
    terrain = make_terrain(4, hardcore=hard)
    if hard:
        from gym_rem2d_amd import synthetic
        batches = [(m, np.asarray(idx, dtype=np.int64)) for m, idx in
                   synthetic.cppn_batches_native(range(args.seed * 10 ** 6, args.seed * 10 ** 6 + args.n), n_proc=4)]
        select = None
    else:
        if args.encoding == "direct":
            pop = DirectPopulation.random(args.n, rng)
        elif args.encoding == "network_arrays":
            pop = NetworkPopulation.random(args.n, rng)
        else:
            pop = LSystemPopulation.random(args.n, rng, max_modules=args.max_modules)
        if args.encoding != "lsystem":
            for _ in range(args.mutations):
                pop.mutate(0.2, 0.2, 0.2, rng)
        batches = pop.compile(0)
    env = BatchedModular2D(hardcore=hard, flags=_lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN)
    env._upload(batches, args.n)
    shapes = sorted({int(w.tile_shape) for w, _ in env.worlds})
    t0 = time.time()
    fit = run_episode(env, max_steps=args.cap).cpu().numpy()
    t_gpu = time.time() - t0
    overflow = sorted(env.last_overflow)
    env.close()
    sample = np.arange(args.n) if args.check >= 1.0 else \
        np.union1d(rng.choice(args.n, int(args.n * args.check), replace=False), np.asarray(overflow, dtype=np.int64))
    xs, ys, polys = terrain.f32()
    ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
    ref = np.full(args.n, np.nan)
    t0 = time.time()
    if hard:   # the lane buckets as uploaded, in chunks of their creatures (Morphology.take keeps the layout)
        want = np.zeros(args.n, dtype=bool)
        want[sample] = True
        for m, idx in batches:
            keep = np.nonzero(want[idx])[0]
            for lo in range(0, len(keep), args.chunk):
                part = keep[lo:lo + args.chunk]
                r = O.batch_run(ot, m.take(part).as_dict(), args.cap, n_threads=os.cpu_count() or 1, flags=O.FLAG_CONTINUOUS)
                ref[idx[part]] = r["fitness"]
    else:
        for lo in range(0, len(sample), args.chunk):
            part = sample[lo:lo + args.chunk]
            for m, idx in pop.select(part).compile(0):
                r = O.batch_run(ot, m.as_dict(), args.cap, n_threads=os.cpu_count() or 1, flags=O.FLAG_CONTINUOUS)
                ref[part[np.asarray(idx)]] = r["fitness"]
    t_cpu = time.time() - t0
    bad = int((fit[sample] != ref[sample]).sum())
    print(json.dumps(dict(encoding=args.encoding, individuals=args.n, cap=args.cap, checked=int(len(sample)), mismatching=bad, wide_fallback=len(overflow),
                          tile_shapes=shapes, gpu_seconds=round(t_gpu, 1), oracle_seconds=round(t_cpu, 1),
                          fitness_mean=float(fit.mean()), fitness_max=float(fit.max()))), flush=True)
    print("SOAK " + ("OK" if bad == 0 else "MISMATCH"))
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
