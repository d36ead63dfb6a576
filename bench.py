#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched stepper on N GPUs of one node.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run, one rank per GPU (RCCL) -- or, started bare, it launches those ranks
itself as a child process (launch_ranks).  One "step" = one Modular2D.step of every
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
  single         config 1: one direct-encoding individual (random.seed(0)), default terrain, from reset; the CPU
                 baseline next to it is the oracle on one thread
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
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (tools/profile_round.sh:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950; newest round first)."""
    for name in ("r02_b_pmc_traffic.json", "r02_a_pmc_traffic.json", "r01_f_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:  # noqa: BLE001
            continue
        # profile keys carry the template arguments ("rem2d_vel4_kernel<1, 1, 1, 4>"); match on the stem
        for k, v in d["kernels"].items():
            if k == kernel_name or k.startswith(kernel_name + "<") or k.startswith(kernel_name.rstrip(">") + ","):
                return float(v["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
    return None, None


def valu_issue(n_groups):
    """VALU wave-instructions per env-step of the default population from the committed SQ-counter pass
    (profiles/r02_b_sq_counters.json: SQ_INSTS_VALU per launch of one step group, tools/profile_round.sh) and the chip's
    measured issue peak (tools/ubench_latency.hip -> profiles/r02_b_ubench_valu_latency.txt: 860 G wave-instructions/s)."""
    path = os.path.join(ROOT, "profiles", "r02_b_sq_counters.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:  # noqa: BLE001
        return None
    per_group = sum(v["SQ_INSTS_VALU"] for k, v in d["kernels"].items()
                    if k.startswith(("rem2d_pre", "rem2d_vel4", "rem2d_post", "rem2d_toi")))
    return {"wave_instructions_per_env_step": per_group * n_groups, "peak_wave_instructions_per_s": 860e9,
            "source": "profiles/r02_b_sq_counters.json (per launch of one of the %d step groups of this command), "
                      "profiles/r02_b_ubench_valu_latency.txt" % n_groups}


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
    import pickle
    import tempfile
    seeds = np.arange(rank * n_envs, (rank + 1) * n_envs)
    # The specs are cached on disk: a later run (in particular one under rocprofv3 --pmc, whose preloaded library
    # has initialised the GPU before python starts -- forking a worker pool from such a process hangs) loads them
    # instead of forking.
    cache = os.path.join(tempfile.gettempdir(), "rem2d_bench_%s_%d_%d.pkl" % (workload, n_envs, rank))
    parts = None
    if os.path.exists(cache):
        try:
            with open(cache, "rb") as f:
                parts = pickle.load(f)
        except Exception:  # noqa: BLE001
            parts = None
    maker = synthetic.cppn_specs if workload == "cppn_hardcore" else synthetic.lsystem_specs
    if parts is None:
        n_proc = max(1, min(8 if n_envs <= 131072 else 32, os.cpu_count() or 1))
        chunks = np.array_split(seeds, n_proc * 8)
        if os.environ.get("REM2D_BENCH_NO_FORK"):
            parts = [maker(c.tolist()) for c in chunks]
        else:
            with mp.get_context("fork").Pool(n_proc) as pool:
                parts = pool.map(maker, [c.tolist() for c in chunks])
        try:
            with open(cache + ".tmp%d" % os.getpid(), "wb") as f:
                pickle.dump(parts, f, protocol=pickle.HIGHEST_PROTOCOL)
            os.replace(cache + ".tmp%d" % os.getpid(), cache)
        except Exception:  # noqa: BLE001
            pass
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


def _repack(m, lanes):
    """A lane bucket re-laid out on `lanes` lanes per creature (empty lanes appended), as one dict for the oracle."""
    K, n = m.lanes, m.n_envs
    out = {"n_envs": n, "lanes": lanes}
    for k, v in m.arrays.items():
        fill = -1 if k == "parent" else 0
        a = np.full((n, lanes), fill, dtype=v.dtype)
        a[:, :K] = v.reshape(n, K)
        out[k] = a.reshape(-1)
    return out


def cpu_baseline(morphs, terrain, flags, settle, window, budget_s=25.0):
    """The oracle (C restatement of the reference's Box2D path, OpenMP over creatures) on a bounded sample of the same
    workload, over the SAME step window the GPU leg times: steps [settle, settle + window) after reset.  The oracle
    has no "continue" call, so the window is the difference of two runs from reset (it is deterministic).  All lane
    buckets go into ONE batch_run (re-laid out on the widest lane count) so that every host thread stays busy."""
    from oracle import oracle as O
    O.build()
    xs, ys, polys = terrain.f32()
    ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
    cores = os.cpu_count() or 1
    total = sum(m.n_envs for m in morphs)
    want = min(total, max(256, 32 * cores))            # >= 32 creatures per host thread
    lanes = max(m.lanes for m in morphs)
    parts = []
    for m in morphs:
        t = min(m.n_envs, max(1, int(round(want * m.n_envs / total))))
        parts.append(_repack(m.take(np.linspace(0, m.n_envs - 1, t).astype(np.int64)), lanes))
    sample = {"n_envs": sum(p["n_envs"] for p in parts), "lanes": lanes}
    for k in parts[0]:
        if k not in ("n_envs", "lanes"):
            sample[k] = np.concatenate([p[k] for p in parts])
    n = sample["n_envs"]

    def timed(steps, threads, sub=None):
        d = sample if sub is None else sub
        t0 = time.perf_counter()
        O.batch_run(ot, d, steps, n_threads=threads, flags=flags)
        return time.perf_counter() - t0
    t_settle = timed(settle, cores)
    # size the window so that the whole leg stays within the budget
    rate_guess = n * max(1, settle) / max(t_settle, 1e-3)
    window = int(max(20, min(max(window, 100), (budget_s * 0.5 * rate_guess / n - settle))))
    t_full = timed(settle + window, cores)
    all_threads = n * window / max(t_full - t_settle, 1e-6)
    # one thread: a slice of the sample, same window
    n1 = max(8, min(n, 32))
    idx = np.linspace(0, n - 1, n1).astype(np.int64)
    lanes_idx = (idx[:, None] * lanes + np.arange(lanes)[None, :]).reshape(-1)
    sub = {"n_envs": n1, "lanes": lanes}
    for k, v in sample.items():
        if k not in ("n_envs", "lanes"):
            sub[k] = v[lanes_idx]
    w1 = max(100, min(window, 200))
    t1a = min(timed(settle, 1, sub) for _ in range(2))
    t1b = min(timed(settle + w1, 1, sub) for _ in range(2))
    one_thread = n1 * w1 / max(t1b - t1a, 1e-3)
    return {"value": all_threads, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "what": "oracle/rem2d_oracle.c (CPU restatement of the reference's Box2D 2.3.x path; pybox2d itself is not installable here)",
            "value_1thread": one_thread,
            "sample": "%d creatures (proportional slice of every lane bucket, one batch on %d lanes), steps [%d, %d) after reset "
                      "-- the window the GPU leg times; OpenMP over creatures on %d threads; 1-thread figure: %d creatures x %d steps"
                      % (n, lanes, settle, settle + window, cores, n1, w1)}


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher around it: start N ranks with torch.distributed.run as a CHILD process
    (this parent has not touched the GPU -- no torch import, no HIP call -- and never execs) and relay its output.
    One rank per GPU over RCCL; with fewer than N GPUs on the node the ranks share GPU 0 and the fitness all-gather
    goes over gloo (a smoke test of the multi-rank path, labelled as such in the JSON)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if line:
        print(line[-1])
    else:
        sys.stdout.write(proc.stdout)
    sys.exit(proc.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="lsystem",
                    choices=["lsystem", "chain8", "chain4", "cppn_hardcore", "generation", "single"])
    ap.add_argument("--envs", type=int, default=None, help="creatures per GPU (default: config size)")
    ap.add_argument("--steps-per-launch", type=int, default=25,
                    help="env-steps per C-ABI step call (one call = that many kernel sequences queued on the stream)")
    ap.add_argument("--settle", type=int, default=60,
                    help="untimed steps right after reset so that creatures have landed (spawn is 2 m up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-groups", type=int, default=None,
                    help="independent halves/thirds of the population stepped on separate streams (default: automatic)")
    ap.add_argument("--pipeline", type=int, default=None, choices=[0, 1, 2, 3],
                    help="3 = tile pipeline pre / rem2d_vel4_kernel / post (the library's default), 0 = fused rem2d_step_multi_kernel")
    ap.add_argument("--discrete", action="store_true",
                    help="b2World(continuousPhysics=False): skip SolveTOI (the default follows pybox2d: continuous)")
    args = ap.parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args)          # does not return
    if env_world is not None and int(env_world) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, env_world))
    if args.step_groups is not None:
        os.environ["REM2D_STEP_GROUPS"] = str(args.step_groups)
    if args.pipeline is not None:
        os.environ["REM2D_PIPELINE"] = str(args.pipeline)  # read once by librem2d at the first step

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    single = args.workload == "single"
    n_envs = args.envs or {"chain4": 4096, "generation": 131072, "single": 1}.get(args.workload, 65536)
    generation = args.workload == "generation"
    if single:
        # config 1 (Demo1_Random_Individual.py:4-36): one direct-encoding individual, random.seed(0), seed-4 terrain, 1000 steps
        from gym_rem2d_amd import synthetic
        from gym_rem2d_amd.compiler import Morphology
        specs = synthetic.direct_specs([0] * n_envs) if n_envs > 1 else synthetic.direct_specs([0])
        morphs = [Morphology.from_specs(specs)]
        workload_desc = ("%d direct-encoding individual(s) (random.seed(0), 5 x mutate(0.5, 0.5, 0.5)), default seed-4 terrain, "
                         "from reset (Demo1_Random_Individual.py path)" % n_envs)
        args.settle = 0
    else:
        morphs, workload_desc = build_population("lsystem" if generation else args.workload, n_envs, rank)
    if generation:
        workload_desc = "one EA generation: whole episodes (evaluate() rule, <= 2500 steps) of " + workload_desc

    import torch
    import torch.distributed as dist
    from gym_rem2d_amd import make_terrain
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import all_gather_fitness

    n_dev = torch.cuda.device_count()   # does not initialise the GPU
    backend = os.environ.get("REM2D_DIST_BACKEND") or ("nccl" if n_dev >= world else "gloo")
    torch.cuda.set_device(local_rank % max(1, n_dev))
    dev = torch.device("cuda", local_rank % max(1, n_dev))
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == "nccl" else torch.device("cpu")

    from gym_rem2d_amd import _lib
    hard = args.workload == "cppn_hardcore"
    flat = not hard and not single
    env = BatchedModular2D(flat=flat, hardcore=hard, seed=4, device=dev,
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
    for w, _ in env.worlds:          # event pairs are created here, outside the timed region
        w.enable_timing(True)
        w.kernel_time_ms()
        w.step_time_ms()
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
    fit = env.fitness
    if world > 1:
        fit = all_gather_fitness(fit.to(cdev), n_envs * world)  # the generation's only collective
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel (HIP events on the launch stream, recorded by the library) ----
    # Every launch covers one step group (all lane buckets of a third of the population in one grid); its time is booked
    # on the group's first world.
    pipeline = int(os.environ.get("REM2D_PIPELINE", "3"))
    ms = launches = 0
    ms_step = n_step = 0
    for w, _ in env.worlds:
        a, b = w.kernel_time_ms()
        c, d = w.step_time_ms()
        w.enable_timing(False)
        ms, launches, ms_step, n_step = ms + a, launches + b, ms_step + c, n_step + d
    merged = len(env.worlds) > 1 and env.merged_launch
    n_groups = max(1, len(env.groups))
    kname = {3: "rem2d_vel4_kernel", 0: "rem2d_step_multi_kernel",
             1: "rem2d_vel_kernel", 2: "rem2d_vel3_kernel"}[pipeline]
    bytes_per_step = float(sum(algorithmic_bytes(m.n_bodies).sum() for m in morphs))
    flops_per_step = float(sum(valu_flops_per_env_step(m.n_bodies).sum() for m in morphs))
    # algorithmic bytes of one launch / its average duration == bytes of all timed launches / their total duration
    avg_ms = ms / max(1, launches)
    achieved = bytes_per_step * args.steps / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    achieved_step = bytes_per_step * args.steps / (ms_step * 1e-3) / 1e9 if ms_step > 0 else None
    valu = flops_per_step * args.steps / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    traffic_bytes, traffic_src = (pmc_traffic(kname) if (args.workload == "lsystem" and not args.discrete and n_envs == 65536)
                                  else (None, None))
    traffic = traffic_bytes / (avg_ms * 1e-3) / 1e9 if (traffic_bytes and avg_ms > 0) else None
    err = int(env.errors().max())
    issue = valu_issue(n_groups) if traffic_src and pipeline == 3 else None
    if issue:
        issue["achieved_wave_instructions_per_s"] = issue["wave_instructions_per_env_step"] * args.steps / dt
        issue["frac"] = issue["achieved_wave_instructions_per_s"] / issue["peak_wave_instructions_per_s"]

    if rank == 0:
        total = n_envs * world
        out = {
            "metric": "env steps/sec (whole node) at %s parallel creatures" % format(total, ",").replace(",", " "),
            "value": total * args.steps / dt,
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
            "config": {"workload": workload_desc, "envs_per_gpu": n_envs,
                       "pipeline": {3: "tile (pre / vel4 / post+toi_scan / toi_heavy)", 0: "fused step kernel + TOI kernels",
                                    1: "split (4-wave velocity kernel)", 2: "split (vel3)"}[pipeline],
                       "kernel_launches_per_env_step_per_group": (4 if not args.discrete else 3) if pipeline == 3 else None,
                       "steps_per_abi_call": spl,
                       "settle_steps": args.settle,
                       "velocity_iterations": 180, "position_iterations": 60, "dt": 0.02,
                       "continuous_physics": not args.discrete,
                       "parallelism": "population sharded over %d rank(s), one per GPU, no per-step collective; one fp64 "
                                      "fitness all-gather (%s)" % (world, backend if world > 1 else "none at 1 rank"),
                       "ranks_share_one_gpu": bool(world > 1 and n_dev < world),
                       "merged_launch": bool(merged), "step_groups": n_groups,
                       "timed_region_s": dt,
                       "solver_errors": err},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "avg_launch_ms": avg_ms, "launches": launches,
                         # the same algorithmic bytes over the device time of ALL kernels of an env-step (pre .. toi_heavy)
                         "achieved_all_step_kernels": achieved_step,
                         "avg_step_sequence_ms": (ms_step / n_step) if n_step else None,
                         # launches of different step groups overlap on the chip, which stretches every launch;
                         # the same bytes over the wall time of the timed region:
                         "achieved_wall": bytes_per_step * args.steps / dt / 1e9,
                         "frac_wall": bytes_per_step * args.steps / dt / 1e9 / 8000.0,
                         "valu_tflops_est": valu, "valu_frac_of_157.3": valu / 157.3,
                         # the roof this path actually runs under: VALU wave-instruction issue (at ~10 of 64 lanes)
                         "valu_issue": issue,
                         "note": "algorithmic bytes B(M,C)=72M+100(M-1)+48C+12, C=2M per env-step (SURVEY 8d), all of them "
                                 "charged to the dominant kernel; the path is bound by the per-wavefront issue interval over "
                                 "the 180+60 Gauss-Seidel sweeps (FP32 VALU; valu_issue.frac of the measured issue "
                                 "peak), not by HBM"},
        }
        if dt < 0.5:
            out["config"]["note"] = "timed region shorter than 0.5 s: expect a few per cent of run-to-run noise"
        if not args.no_cpu_baseline and world == 1:
            from gym_rem2d_amd import make_terrain as _mt
            window = args.steps if not generation else 200
            out["cpu_baseline"] = cpu_baseline(morphs, _mt(4, flat=flat, hardcore=hard), 0 if args.discrete else 1,
                                               args.settle + args.warmup, window)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
