#!/usr/bin/env python3
"""End-to-end cost of one EA generation (clone + mutate, genotype->phenotype on the host pool, batched episode).

    python tools/bench_ea.py [--population 16384] [--generations 2]

One JSON line with the wall-time split.  (bench.py --workload generation times the episode alone, with the
population already compiled.)"""
import argparse
import copy
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--population", type=int, default=16384)
    ap.add_argument("--generations", type=int, default=2)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--encoding", default="lsystem", choices=["lsystem", "direct", "network"],
                    help="with --arrays: which array population (population.LSystemPopulation / DirectPopulation / NetworkPopulation)")
    ap.add_argument("--host-only", action="store_true", help="with --arrays: time select + mutate + compile only (no GPU needed)")
    ap.add_argument("--arrays", action="store_true",
                    help="array-based population (population.LSystemPopulation): vectorised selection / mutation, "
                         "native genotype->phenotype; default: object genomes + fork pool")
    args = ap.parse_args()
    if args.arrays:
        return main_arrays(args)
    from gym_rem2d_amd.ea import Individual, sel_tournament
    from gym_rem2d_amd.encode import encode_population
    random.seed(0)
    t0 = time.time()
    pop = [Individual.random(encoding="lsystem") for _ in range(args.population)]
    t_init = time.time() - t0
    import torch
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import run_episode
    env = BatchedModular2D()
    rows = []
    for gen in range(args.generations):
        t0 = time.time()
        off = [copy.deepcopy(o) for o in sel_tournament(pop, len(pop))] if gen else pop
        for o in off:
            if gen:
                o.mutate(0.01, 0.01, 0.1)
        t_var = time.time() - t0
        t0 = time.time()
        batches = encode_population(off, 8, args.workers)
        t_enc = time.time() - t0
        t0 = time.time()
        env.trees = env.robots = None
        env._upload(batches, len(off))
        fit = run_episode(env).cpu().tolist()
        torch.cuda.synchronize()
        t_gpu = time.time() - t0
        for o, f in zip(off, fit):
            o.fitness = f
        pop = off
        rows.append(dict(generation=gen, select_clone_mutate_s=t_var, encode_s=t_enc, upload_and_episode_s=t_gpu,
                         steps=int(env.steps.max()), best=max(fit)))
    print(json.dumps(dict(metric="wall seconds per EA generation", population=args.population, init_s=t_init,
                          host_workers=args.workers or min(os.cpu_count() or 1, 64), generations=rows)))


def _usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup's CPU quota (bench.host_cores)."""
    import bench
    return bench.host_cores()[0]


def main_arrays(args):
    import numpy as np
    from gym_rem2d_amd.population import DirectPopulation, LSystemPopulation, NetworkPopulation, tournament
    rng = np.random.default_rng(0)
    t0 = time.time()
    pop = {"lsystem": LSystemPopulation, "direct": DirectPopulation, "network": NetworkPopulation}[args.encoding].random(
        args.population, rng)
    t_init = time.time() - t0
    if not args.host_only:
        import torch
        from gym_rem2d_amd.env import BatchedModular2D
        from gym_rem2d_amd.evaluate import run_episode
        from gym_rem2d_amd import _lib
        env = BatchedModular2D(flags=_lib.FLAG_CONTINUOUS | (0 if os.environ.get('REM2D_NO_SKIP') else _lib.FLAG_SKIP_FROZEN))
    rows, fit = [], None
    for gen in range(args.generations):
        t0 = time.time()
        if gen:
            pop = pop.select(tournament(fit, len(pop), rng))
            t_sel = time.time() - t0
            pop.mutate(0.01, 0.01, 0.1, rng)
        else:
            t_sel = 0.0
        t_var = time.time() - t0
        t0 = time.time()
        batches = pop.compile(args.workers or 0)
        t_enc = time.time() - t0
        t0 = time.time()
        if args.host_only:   # a stand-in fitness: the selection / mutation / expression path is what is timed
            fit = np.zeros(len(pop))
            for m, idx in batches:
                fit[np.asarray(idx)] = m.n_bodies
            t_gpu, steps = 0.0, 0
        else:
            env.trees = env.robots = None
            env._upload(batches, len(pop))
            fit = run_episode(env, on_error="penalty").cpu().numpy()
            torch.cuda.synchronize()
            t_gpu, steps = time.time() - t0, int(env.steps.max())
        del batches
        rows.append(dict(generation=gen, select_clone_mutate_s=t_var, of_which_select_s=t_sel, encode_s=t_enc,
                         host_s=t_var + t_enc, upload_and_episode_s=t_gpu, steps=steps, best=float(fit.max())))
    print(json.dumps(dict(metric="wall seconds per EA generation", mode="arrays + native compiler", encoding=args.encoding,
                          population=args.population, init_s=t_init, host_cores=_usable_cores(), generations=rows)))


if __name__ == "__main__":
    main()
