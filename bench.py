#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched stepper on N GPUs of one node.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run, one rank per GPU (RCCL).  One "step" = one Modular2D.step of every
creature of the batch (controller + PID + world.Step(1/50, 180, 60) + reward/done).  Rank 0
prints ONE JSON line.  Weak scaling: every rank steps its own 65 536 creatures; the only
collective is one all-gather of fp32 fitness at the end of the timed region.

Workloads (BASELINE.json configs), all synthetic:
  lsystem  (default, the 65 536-creature config the metric is quoted on): random L-system
           creatures, seeds 0..65535, maxModules=15, flat terrain; creatures are bucketed by
           lane count (2/4/8/16) so that waves are homogeneous
  chain8   65 536 identical 8-module chains (north_star's "8-module creatures")
  chain4   4 096 identical 4-module chains (config 2)
  cppn_hardcore  65 536 network-encoded creatures (synthetic CPPN genome) on the hardcore track (config 4)
  generation     config 5's unit of work: every rank evaluates 131 072 L-system individuals for whole
                 episodes (until every fitness is final, <= 2500 steps) and the ranks all-gather the fitness;
                 `--steps` is ignored, the JSON reports env-steps/s over the executed steps
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The step groups run on three streams next to the caller's; RCCL adds its own.  HIP maps streams onto 4 hardware
# queues by default and streams that share a queue serialise (measured: 20 M instead of 29 M env-steps/s with a
# fourth group stream), so give the runtime 8 queues before it initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402


def algorithmic_bytes(n_bodies):
    """B(M, C) = 72 M + 100 (M-1) + 48 C + 12 bytes per env-step, C = 2 M."""
    M = np.asarray(n_bodies, dtype=np.float64)
    return 72 * M + 100 * np.maximum(M - 1, 0) + 48 * (2 * M) + 12


def valu_flops_per_env_step(n_bodies, vel_iters=180):
    """Counted binary32 operations of the velocity loop per env-step (the part that dominates): per
    iteration a revolute joint costs ~105 flop (motor + 2x2 point solve; ~190 with an active limit) and
    a 2-point contact ~150 flop (two friction rows + block solver).  Estimate with one joint per
    non-root body and one 2-point contact per two bodies."""
    M = np.asarray(n_bodies, dtype=np.float64)
    return vel_iters * (105.0 * np.maximum(M - 1, 0) + 150.0 * 0.5 * M)


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_f_pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this command)."""
    path = os.path.join(ROOT, "profiles", "r01_f_pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        # profile keys carry both template arguments ("rem2d_step_kernel<16, 3>"); match on <K
        stem = kernel_name.rstrip(">")
        for k, v in d["kernels"].items():
            if k == kernel_name or k.startswith(stem + ",") or k.startswith(kernel_name + "<"):
                return float(v["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
        return None, None
    except Exception:  # noqa: BLE001
        return None, None


def build_population(workload, n_envs, rank):
    """Host-side synthetic input, built BEFORE the GPU is initialised (uses a fork pool)."""
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology, lanes_for
    if workload == "chain4":
        m = synthetic.chain_population(n_envs, 4, "left")
        return [m], "%d identical 4-module chain creatures, flat terrain, sinusoidal controller" % n_envs
    if workload == "chain8":
        m = synthetic.chain_population(n_envs, 8, "left")
        return [m], "%d identical 8-module chain creatures, flat terrain, sinusoidal controller" % n_envs
    import multiprocessing as mp
    seeds = np.arange(rank * n_envs, (rank + 1) * n_envs)
    n_proc = max(1, min(8, os.cpu_count() or 1))
    chunks = np.array_split(seeds, n_proc * 8)
    maker = synthetic.cppn_specs if workload == "cppn_hardcore" else synthetic.lsystem_specs
    with mp.get_context("fork").Pool(n_proc) as pool:
        parts = pool.map(maker, [c.tolist() for c in chunks])
    specs = [s for p in parts for s in p]
    groups = {}
    for s in specs:
        groups.setdefault(lanes_for(s.n_bodies), []).append(s)
    # creatures of one wave run in lockstep: sort every bucket by (pipeline period, joint rounds, bodies)
    for k in groups:
        groups[k].sort(key=lambda s: (s.period, max(s.rounds, default=-1), s.n_bodies))
    morphs = [Morphology.from_specs(groups[k], k) for k in sorted(groups)]
    if workload == "cppn_hardcore":
        return morphs, ("%d network-encoded creatures (synthetic feed-forward CPPN genome, seeds %d..%d), hardcore "
                        "terrain (pits/stumps/stairs), bucketed by lane count %s" % (n_envs, seeds[0], seeds[-1], sorted(groups)))
    return morphs, ("%d random L-System creatures (seeds %d..%d, maxModules=15, <=16 bodies), flat terrain, "
                    "bucketed by lane count %s and sorted by joint rounds" % (n_envs, seeds[0], seeds[-1], sorted(groups)))


def cpu_baseline(morphs, terrain, budget_s=12.0, flags=0):
    """The oracle (C restatement, OpenMP over creatures) on a bounded sample of the same workload."""
    from oracle import oracle as O
    O.build()
    xs, ys, _ = terrain.f32()
    xs, ys, polys = terrain.f32()
    ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
    cores = os.cpu_count() or 1
    # sample: a proportional slice of every bucket (>= 4 creatures per host thread), 50 steps from reset;
    # repeated until the time budget is used
    total = sum(m.n_envs for m in morphs)
    want = min(total, max(512, 4 * cores))
    take = [min(m.n_envs, max(1, int(round(want * m.n_envs / total)))) for m in morphs]
    subs = [m.take(np.linspace(0, m.n_envs - 1, t).astype(np.int64)) for m, t in zip(morphs, take)]
    steps, done_steps, t0 = 50, 0, time.time()
    n = sum(s.n_envs for s in subs)
    while True:
        for s in subs:
            O.batch_run(ot, s.as_dict(), steps, n_threads=cores, flags=flags)
        done_steps += steps
        if time.time() - t0 > budget_s or done_steps >= 2000:
            break
    dt = time.time() - t0
    return {"value": n * done_steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d creatures (proportional slice of every lane bucket) x %d steps from reset, "
                      "oracle/rem2d_oracle.c with OpenMP over creatures" % (n, done_steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="lsystem", choices=["lsystem", "chain8", "chain4", "cppn_hardcore", "generation"])
    ap.add_argument("--envs", type=int, default=None, help="creatures per GPU (default: config size)")
    ap.add_argument("--steps-per-launch", type=int, default=10)
    ap.add_argument("--settle", type=int, default=60,
                    help="untimed steps right after reset so that creatures have landed (spawn is 2 m up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-groups", type=int, default=None,
                    help="independent halves/thirds of the population stepped on separate streams (default: automatic)")
    ap.add_argument("--pipeline", type=int, default=None, choices=[0, 1, 2],
                    help="0 = fused rem2d_step_kernel, 1 = split pre/vel/post pipeline (default: the library's default)")
    ap.add_argument("--discrete", action="store_true",
                    help="b2World(continuousPhysics=False): skip SolveTOI (the default follows pybox2d: continuous)")
    args = ap.parse_args()
    if args.step_groups is not None:
        os.environ["REM2D_STEP_GROUPS"] = str(args.step_groups)
    if args.pipeline is not None:
        os.environ["REM2D_PIPELINE"] = str(args.pipeline)  # read once by librem2d at the first step

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_envs = args.envs or {"chain4": 4096, "generation": 131072}.get(args.workload, 65536)
    generation = args.workload == "generation"

    morphs, workload_desc = build_population("lsystem" if generation else args.workload, n_envs, rank)
    if generation:
        workload_desc = "one EA generation: whole episodes (evaluate() rule, <= 2500 steps) of " + workload_desc

    import torch
    import torch.distributed as dist
    from gym_rem2d_amd import make_terrain
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import all_gather_fitness

    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(1, n_dev))
    dev = torch.device("cuda", local_rank % max(1, n_dev))
    backend = os.environ.get("REM2D_DIST_BACKEND", "nccl")  # "gloo" only to smoke-test the multi-rank path on one GPU
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == "nccl" else torch.device("cpu")

    from gym_rem2d_amd import _lib
    hard = args.workload == "cppn_hardcore"
    env = BatchedModular2D(flat=not hard, hardcore=hard, seed=4, device=dev,
                           flags=0 if args.discrete else _lib.FLAG_CONTINUOUS)
    batches, lo = [], 0
    for m in morphs:
        batches.append((m, list(range(lo, lo + m.n_envs))))
        lo += m.n_envs
    env._upload(batches, lo)

    spl = max(1, args.steps_per_launch)

    def run(n):
        left = n
        while left > 0:
            k = min(spl, left)
            env.step(k)
            left -= k

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if generation:
        args.settle = args.warmup = 0
    run(args.settle)
    run(args.warmup)
    for w, _ in env.worlds:
        w.enable_timing(True)
        w.kernel_time_ms()
    sync()
    t0 = time.perf_counter()
    if generation:
        from gym_rem2d_amd.evaluate import EPISODE_CAP
        done_steps = 0
        while done_steps < EPISODE_CAP:
            env.step(100)
            done_steps += 100
            if bool((env.frozen != 0).all()):
                break
        args.steps = done_steps
    else:
        run(args.steps)
    fit = env.fitness.to(torch.float32)
    if world > 1:
        fit = all_gather_fitness(fit.to(cdev), n_envs * world)  # the generation's only collective
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline of the dominant kernel.  Default: the merged step kernel (all lane buckets of a step group in one
    # grid, its time booked on the group's first world); per-bucket launches: the bucket with the most device time.
    times = []
    for w, _ in env.worlds:
        ms, launches = w.kernel_time_ms()
        w.enable_timing(False)
        times.append((ms, launches, w))
    merged = len(env.worlds) > 1 and env.merged_launch
    n_groups = max(1, len(env.groups))
    if merged or len(env.worlds) == 1:
        ms = sum(t[0] for t in times)
        launches = sum(t[1] for t in times)       # every launch covers 1/n_groups of the population
        bytes_per_step = float(sum(algorithmic_bytes(m.n_bodies).sum() for m in morphs))
        flops_per_step = float(sum(valu_flops_per_env_step(m.n_bodies).sum() for m in morphs))
        if max(len(g) for g in env.groups) > 1:
            kname = "rem2d_step_multi_kernel"
        else:
            kname = "rem2d_vel_kernel" if os.environ.get("REM2D_PIPELINE") in ("1", "2") else "rem2d_step_kernel<%d>" % morphs[0].lanes
    else:
        ms, launches, wmax = max(times, key=lambda k: k[0])
        m = morphs[[w for w, _ in env.worlds].index(wmax)]
        bytes_per_step = float(algorithmic_bytes(m.n_bodies).sum())
        flops_per_step = float(valu_flops_per_env_step(m.n_bodies).sum())
        kname = "rem2d_vel_kernel" if os.environ.get("REM2D_PIPELINE") in ("1", "2") else "rem2d_step_kernel<%d>" % m.lanes
    bytes_per_step_all = float(sum(algorithmic_bytes(m.n_bodies).sum() for m in morphs))
    # algorithmic bytes of one launch / its average duration == bytes of all timed launches / their total duration
    avg_ms = ms / max(1, launches)
    achieved = bytes_per_step * args.steps / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    valu = flops_per_step * args.steps / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    traffic_bytes, traffic_src = (pmc_traffic(kname) if (args.workload == "lsystem" and not args.discrete and n_envs == 65536)
                                  else (None, None))
    traffic = traffic_bytes / (avg_ms * 1e-3) / 1e9 if (traffic_bytes and avg_ms > 0) else None
    err = int(env.errors().max())

    if rank == 0:
        out = {
            "metric": "env steps/sec (whole node) at 65 536 parallel creatures",
            "value": n_envs * world * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_desc, "envs_per_gpu": n_envs, "steps_per_launch": spl,
                       "settle_steps": args.settle,
                       "velocity_iterations": 180, "position_iterations": 60, "dt": 0.02,
                       "continuous_physics": not args.discrete,
                       "parallelism": "population sharded over %d GPU(s), no per-step collective" % world,
                       "merged_launch": bool(merged), "step_groups": n_groups,
                       "solver_errors": err},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "avg_launch_ms": avg_ms, "launches": launches,
                         # launches of different step groups overlap on the chip, which stretches every launch;
                         # the same bytes over the wall time of the timed region:
                         "achieved_wall": bytes_per_step_all * args.steps / dt / 1e9,
                         "frac_wall": bytes_per_step_all * args.steps / dt / 1e9 / 8000.0,
                         "valu_tflops_est": valu, "valu_frac_of_157.3": valu / 157.3,
                         "note": "algorithmic bytes B(M,C)=72M+100(M-1)+48C+12, C=2M per env-step (SURVEY 8d); "
                                 "the path is FP32-VALU/latency bound, not HBM bound"},
        }
        if not args.no_cpu_baseline and world == 1:
            from gym_rem2d_amd import make_terrain as _mt
            out["cpu_baseline"] = cpu_baseline(morphs, _mt(4, flat=not hard, hardcore=hard), flags=0 if args.discrete else 1)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
