#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched stepper on N GPUs of one node.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run, one rank per GPU (RCCL) -- or, started bare, it launches those ranks
itself as a child process (launch_ranks).  One "step" = one Modular2D.step of every
creature of the batch (controller + PID + world.Step(1/50, 180, 60) + reward/done).  Rank 0
prints ONE JSON line.  `--scaling weak` (default): every rank steps its own 65 536 creatures; `--scaling strong`:
BASELINE.json's 65 536 creatures are split over the ranks.  Either way the only collective is one all-gather of fp64
fitness at the end of each timed block.  The K steps are timed as a BLOCK (barrier + synchronize on both sides, max over
ranks); blocks are repeated until the timed region is >= --min-time seconds (default 5 s: long enough for an external
utilisation sampler to see the GPU busy) and the MEDIAN block is
reported (`config.blocks_ms` lists every block), so that a cold first block (fresh process on a fresh box) does not
decide the figure.  Kernel-exact timing (HIP events) happens in a pass of its own AFTER the timed region; the secondary
workloads (north_star's 8-module creatures, config 4) are measured in the same process and reported under `secondary`.

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


# the hardware's VALU issue ceiling: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 v_fma_f32 (MI355X_MICROARCH.md "Per-instruction cycle
# constants": `v_fma_f32` (wave64) 2 cyc on a SIMD-32)
HW_VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2
# ... and what fully independent instruction streams measure on the box (tools/ubench_latency.hip mode 4 -> profiles/r06_ubench_issue.txt)
UBENCH_INDEPENDENT = {"source": "profiles/r06_ubench_issue.txt (tools/ubench_latency.bin: eight independent v_fma_f32 accumulators per wavefront)",
                      "wave_instructions_per_s_at_4_waves_per_simd": 856e9, "wave_instructions_per_s_at_8_waves_per_simd": 955e9,
                      "note": "the step train runs at 4 wavefronts per SIMD (128 VGPRs): 856 G/s is what the hardware issues there even for "
                              "fully independent streams; 955 G/s needs 8 wavefronts per SIMD (<= 64 VGPRs); the guide's 1 229 G/s (2 cycles per "
                              "wave64 v_fma_f32 at 2.4 GHz) was not reached by any stream on this box"}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (tools/profile_round.sh:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950; newest round first)."""
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_b_pmc_traffic.json", "r01_f_pmc_traffic.json"):
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


def pmc_traffic_all():
    """HBM bytes of every kernel of one env-step sequence (per launch of one step group), from the same PMC passes."""
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_b_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:  # noqa: BLE001
            continue
        ks = {k.split("<")[0].split("(")[0]: float(v["hbm_bytes_per_launch"]) for k, v in d["kernels"].items()
              if k.startswith(("rem2d_pre", "rem2d_vel4", "rem2d_velpost", "rem2d_post", "rem2d_toi", "rem2d_step_train"))}
        return {"bytes_per_launch_sequence": sum(ks.values()), "by_kernel": ks, "source": os.path.relpath(path, ROOT)}
    return None


def valu_issue(n_groups):
    """VALU wave-instructions per env-step of the default population from the committed SQ-counter pass
    (profiles/r05_sq_counters.json: SQ_INSTS_VALU per launch of one step group, tools/profile_round.sh) and the chip's
    measured issue peak (tools/ubench_latency.hip -> profiles/archive/r02_b_ubench_valu_latency.txt: 860 G wave-instructions/s
    with 8 dependent chains per SIMD).  Only when this run uses the step-group count the counters were collected with."""
    for name in ("r05_sq_counters.json", "r04_sq_counters.json", "r03_sq_counters.json", "r02_b_sq_counters.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:  # noqa: BLE001
            continue
        groups = d.get("step_groups") or 4      # (the round-2 file predates the field: four groups)
        if groups != n_groups:
            return None
        per_group = sum(v["SQ_INSTS_VALU"] for k, v in d["kernels"].items()
                        if k.startswith(("rem2d_pre", "rem2d_vel4", "rem2d_velpost", "rem2d_post", "rem2d_toi", "rem2d_step_train")))
        return {"wave_instructions_per_env_step": per_group * n_groups, "peak_wave_instructions_per_s": 860e9,
                "source": "profiles/%s (per launch of one of the %d step groups of this command), "
                          "profiles/archive/r02_b_ubench_valu_latency.txt" % (name, n_groups)}
    return None


def train_counters():
    """The committed counter passes of the step train (tools/train_profile.sh -> profiles/<round>_step_train_counters.json): per
    env-step of the headline population -- HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction) and VALU
    wave-instructions."""
    for name in ("r06_step_train_counters.json", "r05_step_train_counters.json"):   # (newest round first)
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                d = json.load(f)
            return d, os.path.relpath(path, ROOT)
        except Exception:  # noqa: BLE001
            continue
    return None, None


def build_population(workload, n_envs, first):
    """Host-side synthetic input, first stage, BEFORE the GPU is initialised and before torch is imported (uses a fork
    pool): the genomes of the population.  finish_population() turns them into lane-bucket batches."""
    from gym_rem2d_amd import synthetic
    if workload == "chain4":
        m = synthetic.chain_population(n_envs, 4, "left")
        return ("morphs", [m], "%d identical 4-module chain creatures, flat terrain, sinusoidal controller" % n_envs)
    if workload == "chain8":
        m = synthetic.chain_population(n_envs, 8, "left")
        return ("morphs", [m], "%d identical 8-module chain creatures, flat terrain, sinusoidal controller" % n_envs)
    import tempfile
    seeds = np.arange(first, first + n_envs)
    # Only the genomes (a few dozen numbers per creature, drawn from `random` after random.seed(seed)) are made in Python,
    # by a small fork pool sized to this rank's share of the host cores; tree growth, create_robot, schedule and SoA packing
    # are the native compilers' (rem2d_compile_lsystem / rem2d_compile_network: the same words as the Python compiler,
    # tests/test_bench_host.py).  The genome arrays are cached on disk: a later run (in particular one under rocprofv3
    # --pmc, whose preloaded library has initialised the GPU before python starts -- forking from such a process hangs)
    # loads them instead of forking.
    cppn = workload == "cppn_hardcore"
    maker = synthetic._cppn_genome_chunk if cppn else synthetic._lsystem_genome_chunk
    cache = os.path.join(tempfile.gettempdir(), "rem2d_bench_genomes_%s_%d_%d.npz" % (workload, n_envs, first))
    arrays = None
    if os.path.exists(cache):
        try:
            with np.load(cache) as z:
                arrays = {k: z[k] for k in z.files}
        except Exception:  # noqa: BLE001
            arrays = None
    if arrays is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        n_proc = 1 if os.environ.get("REM2D_BENCH_NO_FORK") else max(1, min(8, host_cores()[0] // max(1, local_world)))
        arrays = synthetic._genome_arrays(maker, seeds, None if cppn else 15, n_proc)
        try:
            tmp = cache + ".tmp%d.npz" % os.getpid()
            np.savez(tmp, **arrays)
            os.replace(tmp, cache)
        except Exception:  # noqa: BLE001
            pass
    return ("genomes", workload, arrays, n_envs, int(seeds[0]), int(seeds[-1]))


def finish_population(prep):
    """Second stage of build_population, AFTER every fork pool of the run is done (the native compilers load librem2d.so,
    which imports torch first -- nothing is forked from a process in that state): genome arrays -> lane-bucket batches."""
    if prep[0] == "morphs":
        return prep[1], prep[2]
    _, workload, arrays, n_envs, seed0, seed1 = prep
    from gym_rem2d_amd import encode, synthetic
    cppn = workload == "cppn_hardcore"
    seeds = (seed0, seed1)
    desc = os.environ.get("REM2D_SORT_DESC", "1") != "0"
    # creatures of one wave run in lockstep: every bucket sorted by (pipeline period, joint rounds, bodies) -- the most
    # complex first, so that their wavefronts (the long ones) are dispatched first
    if cppn:
        wide = encode.compile_network_arrays(arrays, 7, 20, 32)
    else:
        wide = encode.compile_lsystem_arrays(arrays, 8, 15, 16)
    batches = synthetic.bucket_batches(wide, descending=desc)
    morphs = [b for b, _ in batches]
    lanes = [b.lanes for b in morphs]
    if cppn:
        return morphs, ("%d network-encoded creatures (synthetic feed-forward CPPN genome, seeds %d..%d), hardcore "
                        "terrain (pits/stumps/stairs), bucketed by lane count %s" % (n_envs, seeds[0], seeds[-1], lanes))
    return morphs, ("%d random L-System creatures (seeds %d..%d, maxModules=15, <=16 bodies), flat terrain, "
                    "bucketed by lane count %s and sorted by joint rounds" % (n_envs, seeds[0], seeds[-1], lanes))


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


def host_cores():
    """Threads this process may really use: the affinity mask, capped by the cgroup CPU quota (v2 cpu.max, v1 cfs)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:  # noqa: BLE001
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def host_memory_budget():
    """Bytes the CPU-baseline leg may spend on oracle worlds: a quarter of what the cgroup (v2 memory.max minus
    memory.current, v1 limit_in_bytes minus usage_in_bytes) or the machine (MemAvailable) still has, at most 4 GiB."""
    avail = None
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            with open(lim) as f:
                v = f.read().strip()
            if v != "max" and int(v) < (1 << 60):
                with open(cur) as f:
                    avail = max(0, int(v) - int(f.read().strip()))
                break
        except Exception:  # noqa: BLE001
            continue
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    m = int(line.split()[1]) * 1024
                    avail = m if avail is None else min(avail, m)
    except Exception:  # noqa: BLE001
        pass
    if avail is None:
        avail = 4 << 30
    return int(min(avail // 4, 4 << 30))


ORACLE_WORLD_BYTES = 600 * 1024   # sizeof(o_world) with O_MAX_BODY_CONTACTS = 32, rounded up


def cpu_baseline(morphs, terrain, flags, settle, window, budget_s=20.0):
    """The oracle (C restatement of the reference's Box2D path, OpenMP over creatures) on a bounded sample of the same
    workload, over the SAME step window the GPU leg times: steps [settle, settle + window) after reset, as ONE continuous
    wall-clock window (rem2d_oracle_batch_window: the settle steps are untimed, the worlds are kept).  All lane buckets
    go into one batch (re-laid out on the widest lane count) so that every host thread stays busy."""
    from oracle import oracle as O
    O.build()
    xs, ys, polys = terrain.f32()
    ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
    cores, quota = host_cores()
    total = sum(m.n_envs for m in morphs)
    lanes = max(m.lanes for m in morphs)

    def sample_of(want):
        parts = []
        for m in morphs:
            t = min(m.n_envs, max(1, int(round(want * m.n_envs / total))))
            parts.append(_repack(m.take(np.linspace(0, m.n_envs - 1, t).astype(np.int64)), lanes))
        smp = {"n_envs": sum(p["n_envs"] for p in parts), "lanes": lanes}
        for k in parts[0]:
            if k not in ("n_envs", "lanes"):
                smp[k] = np.concatenate([p[k] for p in parts])
        return smp
    # one thread first (a small slice): it also sizes the all-thread leg to the budget
    sub = sample_of(min(total, 32))
    n1 = sub["n_envs"]
    w1 = max(50, min(window, 200))
    t1, _ = O.batch_window(ot, sub, settle, w1, n_threads=1, flags=flags)
    one_thread = n1 * w1 / max(t1, 1e-6)
    # all threads: about budget_s seconds of CPU wall time in all (settle + window), >= 48 creatures per thread (creature
    # costs vary a lot), at most 8192 worlds (~0.4 MB each); the window is the GPU leg's, stretched to 100..400 steps
    window = int(max(100, min(window, 400)))
    est_rate = one_thread * cores * 0.7
    want = int(budget_s * est_rate / (settle + window))
    mem_cap = max(64, host_memory_budget() // ORACLE_WORLD_BYTES)   # (the worlds of the window are alive at once)
    sample = sample_of(min(total, mem_cap, max(256, 48 * cores, min(want, 8192))))
    n = sample["n_envs"]
    # (the sample is capped: a longer window of the same creatures fills the budget instead -- about 10-20 s of CPU work)
    window = int(max(window, min(budget_s * est_rate / n - settle, 800)))
    tw, _ = O.batch_window(ot, sample, settle, window, n_threads=cores, flags=flags)
    all_threads = n * window / max(tw, 1e-6)
    return {"value": all_threads, "unit": "env-steps/s", "cores": cores, "kind": "port", "is_oracle": True,
            "cpu_count": os.cpu_count(), "cgroup_cpu_quota": quota,
            "what": "oracle/rem2d_oracle.c -- this repository's test oracle, a CPU restatement of the reference's Box2D 2.3.x "
                    "path (pybox2d itself is not installable here), OpenMP over creatures",
            "value_1thread": one_thread,
            "sample": "%d creatures (proportional slice of every lane bucket, one batch on %d lanes), steps [%d, %d) after reset "
                      "as one continuous timed window of %.2f s (the settle steps untimed) -- the window the GPU leg times; %d "
                      "threads (affinity mask / cgroup quota); 1-thread figure: %d creatures x %d steps"
                      % (n, lanes, settle, settle + window, tw, cores, n1, w1)}


def _reference_worker(job):
    """One worker of reference_baseline (a forked process, like a worker of the reference's multiprocessing pool): the reference's
    own env over real Box2D for a slice of config 3's seeds -- reset, `settle` untimed steps, `window` timed steps per creature."""
    reference, seeds, settle, window = job
    import random
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import capture_golden as cg
    cg.REF = reference
    if reference not in sys.path:
        sys.path.insert(0, reference)
    try:
        import gym  # noqa: F401
    except ImportError:
        cg.install_gym_stub()
    try:
        import neat  # noqa: F401
    except ImportError:
        cg.install_neat_stub()
    import matplotlib
    matplotlib.use("Agg")
    cg.alias_case_insensitive()
    from Encodings import lsystem as ls
    from gym_rem2D.morph import simple_module, circular_module
    from gym_rem2D.envs import Modular2DEnv as M
    M.MAX_PERTURBANCE_TERRAIN = 0   # config 3: flat terrain
    steps, spent = 0, 0.0
    env = M.Modular2D()
    for seed in seeds:
        random.seed(int(seed))
        ml = [simple_module.Standard2D() for _ in range(4)] + [circular_module.Circular2D() for _ in range(4)]
        g = ls.LSystem(ml)
        g.maxModules = 15
        tree = g.create(8)
        env.seed(4)
        env.reset(tree=tree, module_list=ml)     # REM2D_main.py:357-359
        action = np.ones(4)
        for _ in range(settle):
            env.step(action)
        t0 = time.perf_counter()
        for _ in range(window):                  # REM2D_main.py:362-368: the body of evaluate()'s loop, no early exit
            env.step(action)
        spent += time.perf_counter() - t0
        steps += window
    return steps, spent


def reference_baseline(settle, window, budget_s=20.0):
    """SURVEY 8d's timing plan: `try: import Box2D` -> the REFERENCE's own Modular2D.step over the real engine, one env per host core
    the way REM2D_main.py:256-267's pool.map runs evaluate() (:350-378), on the first creatures of config 3.  Needs the wheel
    (requirements.txt:1 of the reference: Box2D==2.3.10) AND a checkout of the reference named by REM2D_REFERENCE (nothing is
    read from /root/reference by default: it does not exist on the GPU box).  Returns None when either is missing -- which is the
    case in this image and on the GPU box, so `cpu_baseline.kind` is "port" there."""
    try:
        import Box2D  # noqa: F401
    except ImportError:
        return None
    reference = os.environ.get("REM2D_REFERENCE", "")
    if not reference or not os.path.isdir(os.path.join(reference, "gym_rem2D")):
        return None
    import multiprocessing as mp
    cores, quota = host_cores()
    # size: one creature first (its rate sizes the sample to the budget)
    n1, t1 = _reference_worker((reference, [0, 1], settle, min(window, 50)))
    rate1 = n1 / max(t1, 1e-6)
    per_core = max(2, min(64, int(budget_s * rate1 / (settle + window))))
    jobs = [(reference, list(range(c * per_core, (c + 1) * per_core)), settle, window) for c in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        parts = pool.map(_reference_worker, jobs)
    wall = time.perf_counter() - t0
    steps = sum(p[0] for p in parts)
    busy = max(p[1] for p in parts)          # the timed windows run side by side: the slowest worker's is the wall time of the window
    import Box2D as B
    return {"value": steps / max(busy, 1e-6), "unit": "env-steps/s", "cores": cores, "kind": "reference", "is_oracle": False,
            "cpu_count": os.cpu_count(), "cgroup_cpu_quota": quota,
            "what": "the reference's own gym_rem2D.envs.Modular2DEnv.Modular2D.step over pybox2d %s, one env per worker process "
                    "(REM2D_main.py:256-267 pool.map)" % getattr(B, "__version__", "?"),
            "value_1thread": rate1,
            "sample": "%d creatures (config 3's seeds 0..%d, %d per worker), steps [%d, %d) after reset timed per creature, %d "
                      "worker processes; pool wall time incl. resets and settle steps %.1f s"
                      % (cores * per_core, cores * per_core - 1, per_core, settle, settle + window, cores, wall)}


# Context for the baseline (measured by the survey in the build container with a recording world in
# place of Box2D): what the reference's step() costs in Python alone, before any physics.
REFERENCE_PYTHON_OVERHEAD = {"step_us_color_control_on": 2287, "step_us_color_control_off": 8, "reset_ms": 2.9,
                             "source": "SURVEY.md 6 (Modular2DEnv.py:624-628 COLOR_CONTROL = a matplotlib colour-map lookup per "
                                       "node and step; stub world.Step, 8-module L-system creature, one core) -- Python only, "
                                       "Box2D not included"}


def n1_reference():
    """The committed single-GPU line of the driver's command that a multi-GPU line is compared with (newest round first)."""
    for name in ("r06_bench_default.json", "r05_bench_step_train_default.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.loads(f.read().strip().splitlines()[-1])
            if d.get("n_gpus") == 1 and d.get("creatures_total") == 65536:
                return {"value": d["value"], "source": "profiles/" + name}
        except Exception:  # noqa: BLE001
            continue
    return None


def scaling_claim(world, strong, total, per_gpu, value):
    ref = n1_reference()
    out = {"north_star": ">= 6x at 8 GPUs", "judged_on": "weak scaling: 65 536 creatures per GPU, creatures_total = N x 65 536 (config 5's "
                                                         "sharding: independent individuals, one fitness all-gather)",
           "this_line": "strong" if strong else "weak", "creatures_total": total, "creatures_per_gpu": per_gpu,
           "n1_reference": ref, "vs_n1": (value / ref["value"]) if ref else None,
           "expected": "weak: ~N x the single-GPU figure (no per-step collective; one 8 B/individual all-gather per block); strong "
                       "(65 536 creatures in all): ~1.2x at 8 GPUs -- a GPU steps an eighth of the population in 0.83 ms against 0.97 ms "
                       "for all of it, the step train is as long as its heaviest creature's own chain (DESIGN.md 7)"}
    return out


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


METRIC = "env steps/sec (whole node) at 65 536 parallel creatures; 1/2/4/8-GPU scaling"   # BASELINE.json's, verbatim


def make_env(morphs, dev, hard, flat, discrete, wide=False):
    from gym_rem2d_amd import _lib
    from gym_rem2d_amd.env import BatchedModular2D
    env = BatchedModular2D(flat=flat, hardcore=hard, seed=4, device=dev, flags=0 if discrete else _lib.FLAG_CONTINUOUS, wide=wide)
    batches, lo = [], 0
    for m in morphs:
        batches.append((m, list(range(lo, lo + m.n_envs))))
        lo += m.n_envs
    env._upload(batches, lo)
    return env


def stepper(env, spl):
    def run(n):
        left = n
        while left > 0:
            k = min(spl, left)
            env.step(k)
            left -= k
    return run


def timed_blocks(run_block, steps, min_time, max_blocks, sync, reduce_max):
    """Repeat [sync, K steps (+ what belongs to a block), sync] until the blocks add up to min_time seconds (at least one,
    at most max_blocks).  Every block is exactly K steps between a barrier + synchronize on both sides; its time is the
    max over ranks.  Returns the list of block times in seconds."""
    blocks = []
    while True:
        sync()
        t0 = time.perf_counter()
        run_block(steps)
        sync()
        blocks.append(reduce_max(time.perf_counter() - t0))
        if sum(blocks) >= min_time or len(blocks) >= max_blocks:
            return blocks


def main():
    t_start = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="lsystem",
                    choices=["lsystem", "chain8", "chain4", "cppn_hardcore", "generation", "single"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank steps its own --envs creatures; strong: --envs creatures in all, split over the ranks")
    ap.add_argument("--envs", type=int, default=None, help="creatures per GPU (weak) / in all (strong); default: config size")
    ap.add_argument("--steps-per-launch", type=int, default=50,
                    help="env-steps per C-ABI step call (one call = that many kernel sequences queued on the streams)")
    ap.add_argument("--settle", type=int, default=60,
                    help="untimed steps right after reset so that creatures have landed (spawn is 2 m up)")
    ap.add_argument("--min-time", type=float, default=5.0,
                    help="repeat the --steps block until the timed region is at least this long (seconds); the median block is reported")
    ap.add_argument("--max-blocks", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (chain8, cppn_hardcore) measured after the headline one")
    ap.add_argument("--step-groups", type=int, default=None,
                    help="independent parts of the population stepped on separate streams (default: automatic)")
    ap.add_argument("--graph", type=int, default=None, choices=[0, 1], help="replay every step call as a hipGraph (REM2D_GRAPH)")
    ap.add_argument("--pipeline", type=int, default=None, choices=[0, 3],
                    help="3 = tile pipeline pre / velocity tiles / post (the library's default), 0 = fused rem2d_step_multi_kernel "
                         "(launch option `pipeline` of every world, rem2d_world_set_option)")
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
        os.environ["REM2D_PIPELINE"] = str(args.pipeline)  # gym_rem2d_amd._lib.env_options -> rem2d_world_set_option per world
    if args.graph is not None:
        os.environ["REM2D_GRAPH"] = str(args.graph)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    single = args.workload == "single"
    generation = args.workload == "generation"
    n_config = args.envs or {"chain4": 4096, "generation": 131072, "single": 1}.get(args.workload, 65536)
    strong = args.scaling == "strong" and world > 1
    if strong:
        # BASELINE's population cut into contiguous shards: rank r owns creatures [r * ceil(n / W), ...) -- evaluate.shard_range
        per = -(-n_config // world)
        first, n_envs = rank * per, max(0, min(n_config, (rank + 1) * per) - rank * per)
        if n_envs == 0:
            sys.exit("bench.py: more ranks than creatures")
    else:
        first, n_envs = rank * n_config, n_config
    total = n_config if strong else n_config * world
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
        prep = build_population("lsystem" if generation else args.workload, n_envs, first)
    # the secondary workloads' host-side input, also before the GPU is initialised (fork pool)
    secondary_in = {}
    if args.workload == "lsystem" and world == 1 and not args.no_secondary and n_envs == 65536 and not args.discrete:
        for wl in ("chain8", "cppn_hardcore"):
            secondary_in[wl] = build_population(wl, 65536, 0)
    t_genomes = time.perf_counter()
    if not single:
        morphs, workload_desc = finish_population(prep)
    secondary_in = {wl: finish_population(p) for wl, p in secondary_in.items()}
    if generation:
        workload_desc = "one EA generation: whole episodes (evaluate() rule, <= 2500 steps) of " + workload_desc
    if world > 1:   # (the population `value` is quoted on, in the workload's own name)
        workload_desc = ("%d creatures in all = %d ranks x %d (%s scaling); this rank: " % (total, world, n_envs, "strong" if strong else "weak")) \
            + workload_desc

    import torch
    import torch.distributed as dist
    from gym_rem2d_amd import make_terrain
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

    hard = args.workload == "cppn_hardcore"
    flat = not hard and not single
    t_compiled = time.perf_counter()
    env = make_env(morphs, dev, hard, flat, args.discrete)
    spl = max(1, args.steps_per_launch)
    run = stepper(env, spl)
    per_rank = -(-total // world)       # all_gather_fitness pads every rank's shard to this

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def block(n):
        run(n)
        fit = env.fitness
        if world > 1:
            all_gather_fitness(fit.to(cdev), per_rank * world)   # the job's only collective

    # predicted cost of every rank's shard (bodies to step; outside any timed region -- measurement plumbing, not the job's
    # collective): the line shows how even the cut is
    my_cost = int(sum(int(m.n_bodies.sum()) for m in morphs)) if not single else 0
    shard_cost = [my_cost]
    if world > 1:   # (the same kind of collective as the fitness all-gather, on the same device: nothing new for RCCL to do)
        mine = torch.tensor([my_cost], dtype=torch.int64, device=cdev)
        every = torch.empty(world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(every, mine)
        shard_cost = [int(v) for v in every.cpu().tolist()]
    if generation:
        args.settle = args.warmup = 0
    run(args.settle)
    run(args.warmup)
    sync()
    t_first_block = time.perf_counter()   # every rank is here: the timed blocks start now
    startup = {"genomes_s": round(t_genomes - t_start, 2), "native_compile_and_rendezvous_s": round(t_compiled - t_genomes, 2),
               "upload_settle_warmup_s": round(t_first_block - t_compiled, 2), "to_first_block_s": round(t_first_block - t_start, 2),
               "host_cores": host_cores()[0]}
    if generation:
        from gym_rem2d_amd.evaluate import EPISODE_CAP
        sync()
        t0 = time.perf_counter()
        done_steps = 0
        while done_steps < EPISODE_CAP:
            env.step(100)
            done_steps += 100
            if bool((env.frozen != 0).all()):
                break
        args.steps = done_steps
        fit = env.fitness
        if world > 1:
            all_gather_fitness(fit.to(cdev), per_rank * world)
        sync()
        blocks = [reduce_max(time.perf_counter() - t0)]
    else:
        blocks = timed_blocks(block, args.steps, args.min_time, args.max_blocks, sync, reduce_max)
    dt = float(np.median(blocks))
    err = int(env.errors().max())

    # ---- roofline of the dominant kernel: a pass of its own AFTER the timed region, with HIP events recorded by the
    # library on the launch streams (one more block of K steps of the same population, continuing where the timed
    # region stopped).  Every launch covers one step group (all lane buckets of that part of the population in one
    # grid); its time is booked on the group's first world. ----
    pipeline = env.worlds[0][0].get_option("pipeline")   # (the launch option the library really runs with)
    ms = launches = 0
    ms_step = n_step = 0
    timing_steps = 0
    if not generation:
        firsts = [env.worlds[g[0]][0] for g in env.groups] if env.groups else [env.worlds[0][0]]
        timing_steps = min(args.steps, 200)
        for w in firsts:          # event pairs are created here (as many as this pass records); only each group's first world records
            w.enable_timing(timing_steps + 8)
            w.kernel_time_ms()
            w.step_time_ms()
        sync()
        run(timing_steps)
        sync()
        for w in firsts:
            a, b = w.kernel_time_ms()
            c, d = w.step_time_ms()
            w.enable_timing(False)
            ms, launches, ms_step, n_step = ms + a, launches + b, ms_step + c, n_step + d
    merged = len(env.worlds) > 1 and env.merged_launch
    n_groups = max(1, len(env.groups))
    kname = {3: "rem2d_vel4_kernel", 0: "rem2d_step_multi_kernel"}[pipeline]
    fused_velpost = False
    if pipeline == 3:
        # (the library's answer: velocity tiles and position iterations of a 64-lane block in ONE launch -- then that launch,
        # rem2d_velpost_kernel, is the dominant kernel and what the HIP events above timed)
        launch_shape, fused_velpost = env.launch_info()
        if fused_velpost == 2:     # the step train: all steps of a call and all phases of a step in one launch
            kname = "rem2d_step_train128_kernel" if launch_shape in (1, 4) else "rem2d_step_train_kernel"
        elif fused_velpost:
            kname = "rem2d_velpost_kernel"
    bytes_per_step = float(sum(algorithmic_bytes(m.n_bodies).sum() for m in morphs))
    flops_per_step = float(sum(valu_flops_per_env_step(m.n_bodies).sum() for m in morphs))
    # algorithmic bytes of one launch / its average duration == bytes of all timed launches / their total duration
    avg_ms = ms / max(1, launches)
    achieved = bytes_per_step * timing_steps / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    achieved_step = bytes_per_step * timing_steps / (ms_step * 1e-3) / 1e9 if ms_step > 0 else None
    valu = flops_per_step * timing_steps / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    headline = args.workload == "lsystem" and not args.discrete and n_envs == 65536
    if headline and fused_velpost == 2:
        # the step train: one kernel holds the whole step; its counters are booked per env-step of the population (a launch
        # covers as many steps as the ABI call asks for), so traffic = bytes per env-step x the steps the timing pass ran / its time
        tc, traffic_src = train_counters()
        pe = tc["per_env_step"] if tc else {}
        traffic_bytes = pe.get("hbm_bytes")
        traffic = traffic_bytes * timing_steps / (ms * 1e-3) / 1e9 if (traffic_bytes and ms > 0) else None
        traffic_all = {"bytes_per_env_step": traffic_bytes, "bytes_per_env_step_raw": pe.get("hbm_bytes_raw"),
                       "algorithmic_bytes_per_env_step": bytes_per_step, "source": traffic_src} if traffic_bytes else None
        issue = None
        if pe.get("SQ_INSTS_VALU") and tc.get("step_groups") == n_groups:
            issue = {"wave_instructions_per_env_step": pe["SQ_INSTS_VALU"], "peak_wave_instructions_per_s": 860e9,
                     "active_lanes_per_valu_inst": tc.get("active_lanes_per_valu_inst"),
                     "scalar_instructions_per_env_step": pe.get("SQ_INSTS_SALU"),
                     "wait_frac_of_wave_cycles": (pe["SQ_WAIT_ANY"] / pe["SQ_WAVE_CYCLES"]) if pe.get("SQ_WAVE_CYCLES") else None,
                     "source": "%s (per env-step of this population), profiles/archive/r02_b_ubench_valu_latency.txt" % traffic_src}
    else:
        traffic_bytes, traffic_src = pmc_traffic(kname) if headline else (None, None)
        traffic = traffic_bytes / (avg_ms * 1e-3) / 1e9 if (traffic_bytes and avg_ms > 0) else None
        traffic_all = pmc_traffic_all() if headline else None
        issue = valu_issue(n_groups) if headline and pipeline == 3 else None
    if issue:
        issue["achieved_wave_instructions_per_s"] = issue["wave_instructions_per_env_step"] * args.steps / dt
        issue["frac"] = issue["achieved_wave_instructions_per_s"] / issue["peak_wave_instructions_per_s"]
        # two ceilings, both printed: `peak_wave_instructions_per_s` / `frac` = what DEPENDENT chains reach on this
        # chip (tools/ubench_latency.hip, 8 chains per SIMD) -- the shape of a Gauss-Seidel sweep; `hw_*` = the hardware's VALU issue
        # rate, wave64 v_fma_f32 at 2 cycles on each of 1024 SIMD-32s at 2.4 GHz (MI355X_MICROARCH.md, cycle constants), which only
        # independent instruction streams reach (measured: UBENCH_INDEPENDENT below)
        issue["hw_peak_wave_instructions_per_s"] = HW_VALU_ISSUE_PEAK
        issue["hw_frac"] = issue["achieved_wave_instructions_per_s"] / HW_VALU_ISSUE_PEAK
        issue["ubench_independent_streams"] = UBENCH_INDEPENDENT

    # ---- secondary workloads, same process: north_star's "8-module creatures" and config 4 ----
    secondary = None
    if secondary_in and rank == 0:
        secondary = {}
        env.close()
        for wl, (m2, desc2) in secondary_in.items():
            hard2 = wl == "cppn_hardcore"
            env2 = make_env(m2, dev, hard2, not hard2, False)
            run2 = stepper(env2, spl)
            run2(args.settle)
            run2(20)
            b2 = timed_blocks(run2, 100, 0.5, 50, sync, reduce_max)
            d2 = float(np.median(b2))
            secondary[wl] = {"value": 65536 * 100 / d2, "unit": "env-steps/s", "ms_per_step": d2 / 100 * 1e3,
                             "workload": desc2, "steps_per_block": 100, "blocks_ms": [round(x * 1e3, 3) for x in b2],
                             "timed_region_s": float(sum(b2)), "step_groups": max(1, len(env2.groups)),
                             "tile_shape": env2._tile_shape_used, "solver_errors": int(env2.errors().max())}
            env2.close()
        # the headline workload once more in the labelled TOLERANCE MODE (librem2d_fma.so: the same source with
        # -ffp-contract=fast; NOT bit-exact -- tests/test_parity_gpu.py::test_fma_tolerance_mode_transition_parity states what it
        # keeps: integer state in >= 99.98 % of the transitions, poses within 1e-5 + 1e-4 |x| in >= 99.95 %).  A secondary figure:
        # the strict build is the headline and the only thing the digests guard.
        try:
            env3 = make_env(morphs, dev, hard, flat, False, wide="fma")
            run3 = stepper(env3, spl)
            run3(args.settle)
            run3(20)
            b3 = timed_blocks(run3, 100, 0.5, 50, sync, reduce_max)
            d3 = float(np.median(b3))
            from gym_rem2d_amd import _lib as _l3
            secondary["fma_tolerance_mode"] = {
                "value": total * 100 / d3, "unit": "env-steps/s", "ms_per_step": d3 / 100 * 1e3,
                "workload": workload_desc + " -- TOLERANCE MODE (-ffp-contract=fast build, not bit-exact; validated transition by "
                                            "transition against the strict build)", "build_id": _l3.build_id("fma"),
                "steps_per_block": 100, "blocks_ms": [round(x * 1e3, 3) for x in b3], "timed_region_s": float(sum(b3)),
                "step_groups": max(1, len(env3.groups)), "tile_shape": env3._tile_shape_used,
                "solver_errors": int(env3.errors().max())}
            env3.close()
        except Exception as e:   # (a missing librem2d_fma.so must not take the headline down with it)
            secondary["fma_tolerance_mode"] = {"error": str(e)}

    if rank == 0:
        from gym_rem2d_amd import _lib as _l
        _build_id = _l.build_id()
        out = {
            "metric": METRIC,
            "value": total * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "world_size": world,                                  # ranks that took part in the job's collective
            "backend": backend if world > 1 else None,            # "nccl" = RCCL over xGMI, "gloo" = ranks sharing one GPU
            "build_id": _build_id,
            "creatures_total": total,   # the population `value` is quoted on (weak scaling: 65 536 PER GPU, see `scaling`)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            # north_star's ">= 6x scaling at 8 GPUs" is judged on WEAK scaling (65 536 creatures per GPU = config 5's sharding);
            # the line carries the single-GPU figure it is compared with (the driver computes the efficiency itself from its own runs)
            "scaling_claim": scaling_claim(world, strong, total, n_envs, total * args.steps / dt)
            if args.workload == "lsystem" and not args.discrete and not args.envs else None,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_desc, "creatures_total": total, "envs_per_gpu": n_envs,
                       "pipeline": ("step train (one launch per call: a workgroup per (step, 64-lane block) runs pre + vel4 + post + the TOI solve "
                                    "of its own bodies, block-steps handed over through flags inside an XCD)" if fused_velpost == 2 else
                                    "tile (pre / velpost = vel4 + post + toi_scan / toi_heavy)" if fused_velpost else
                                    {3: "tile (pre / vel4 / post+toi_scan / toi_heavy)", 0: "fused step kernel + TOI kernels"}[pipeline]),
                       "kernel_launches_per_env_step_per_group": (None if fused_velpost == 2 else
                                                                  (4 if not args.discrete else 3) - (1 if fused_velpost else 0))
                       if pipeline == 3 else None,
                       # (the step train: ONE launch per ABI call -- plus one per re-ordering of the creature order inside it)
                       "kernel_launches_per_abi_call": 1 if fused_velpost == 2 else None,
                       "steps_per_abi_call": spl,
                       "settle_steps": args.settle,
                       "velocity_iterations": 180, "position_iterations": 60, "dt": 0.02,
                       "continuous_physics": not args.discrete,
                       "parallelism": "population sharded over %d rank(s), one per GPU, no per-step collective; one fp64 "
                                      "fitness all-gather per block (%s)" % (world, backend if world > 1 else "none at 1 rank"),
                       "ranks_share_one_gpu": bool(world > 1 and n_dev < world),
                       "shard_cost_bodies": shard_cost,   # per rank: bodies in its shard (the static cost key of evaluate.shard_balanced)
                       "merged_launch": bool(merged), "step_groups": n_groups, "launch": kname, "hip_graph": bool(env.use_graph),
                       # every block = exactly `steps` env-steps between barrier + synchronize; value = median block
                       "blocks": len(blocks), "blocks_ms": [round(x * 1e3, 3) for x in blocks],
                       "block_ms_median": dt * 1e3, "block_ms_first": blocks[0] * 1e3, "block_ms_min": min(blocks) * 1e3,
                       "timed_region_s": float(sum(blocks)),
                       "startup": startup,   # wall seconds of this rank from process start to the first timed block
                       "solver_errors": err},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         # HBM bytes of ALL kernels of an env-step sequence of one step group (pre .. toi_heavy), same PMC passes
                         "traffic_all_kernels": traffic_all,
                         "kernel": kname, "avg_launch_ms": avg_ms, "launches": launches,
                         "timing_pass": "%d env-steps after the timed region, HIP events on the launch streams" % timing_steps,
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
                         # (top level as well: the two figures that say how the chip is used -- of 64 lanes, and of a resident wavefront's cycles)
                         "active_lanes_per_valu_inst": issue.get("active_lanes_per_valu_inst") if issue else None,
                         "wait_frac_of_wave_cycles": issue.get("wait_frac_of_wave_cycles") if issue else None,
                         "note": "algorithmic bytes B(M,C)=72M+100(M-1)+48C+12, C=2M per env-step (SURVEY 8d), all of them "
                                 "charged to the dominant kernel; the path is bound by the per-wavefront issue interval over "
                                 "the 180+60 Gauss-Seidel sweeps (FP32 VALU; valu_issue.frac of the issue "
                                 "peak), not by HBM"},
        }
        if secondary is not None:
            out["secondary"] = secondary
        if not args.no_cpu_baseline and world == 1:
            from gym_rem2d_amd import make_terrain as _mt
            window = args.steps if not generation else 200
            # the real reference if the wheel and a checkout are there (kind "reference"), else the port
            ref = reference_baseline(args.settle + args.warmup, int(max(100, min(window, 400)))) \
                if (args.workload == "lsystem" and not args.discrete) else None
            out["cpu_baseline"] = ref if ref is not None else \
                cpu_baseline(morphs, _mt(4, flat=flat, hardcore=hard), 0 if args.discrete else 1, args.settle + args.warmup, window)
            out["cpu_baseline"]["reference_python_overhead"] = REFERENCE_PYTHON_OVERHEAD
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
