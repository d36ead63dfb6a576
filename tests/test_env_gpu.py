"""Reference-shaped env API on the GPU path (pytest -m gpu)."""
import copy
import random

import os

import numpy as np
import pytest

from conftest import oracle_terrain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def need_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()


class _Individual:
    def __init__(self, genome):
        self.genome, self.tree_depth, self.fitness = genome, 8, 0


def _individual(seed, enc="direct"):
    from gym_rem2d_amd import get_module_list
    from gym_rem2d_amd.encodings import DirectEncoding, LSystem
    random.seed(seed)
    ml = get_module_list()
    return _Individual(DirectEncoding(ml) if enc == "direct" else LSystem(ml))


def test_modular2d_facade_matches_oracle(need_gpu, oracle, rough_terrain):
    from gym_rem2d_amd import Morphology, build_creature
    from gym_rem2d_amd.env import Modular2D
    ind = _individual(0)
    env = Modular2D()
    assert env.seed(4) == [4]
    tree = ind.genome.create(8)
    assert env.reset(tree=tree, module_list=ind.genome.moduleList) is None
    assert all(not n.expressed for n in tree.getNodes())         # the env works on a deep copy
    assert all(n.expressed for n in env.tree_morphology.nodes if n.component is not None)
    assert len(env.robot.joints) == len(env.robot.components) - 1
    t2 = copy.deepcopy(tree)
    spec, _, _ = build_creature(t2.getNodes(), ind.genome.moduleList)
    ow = oracle.World.from_morph(oracle_terrain(oracle, rough_terrain), Morphology.from_specs([spec]).as_dict(), 0,
                                 flags=oracle.FLAG_CONTINUOUS)  # pybox2d default: continuousPhysics on
    for k in range(150):
        obs, reward, done, info = env.step(np.ones(4))
        r, d = ow.env_step()
        assert obs == 0 and info == 0
        assert reward == r and bool(done) == bool(d)
        assert env.wod.position == pytest.approx(0.04 * (k + 1))
        if d:
            break
    x = env.robot.components[0].position[0]
    assert x == float(ow.bodies()[0][0])
    env.close()
    env2 = Modular2D()
    env2.reset()
    with pytest.raises(Exception, match="no tree_morphology"):
        env2.step(None)


def test_evaluate_and_population(need_gpu, oracle, rough_terrain):
    from gym_rem2d_amd import Morphology, build_creature
    from gym_rem2d_amd.evaluate import evaluate, evaluate_population
    inds = [_individual(s, "lsystem" if s % 2 else "direct") for s in range(12)]
    specs = []
    for ind in inds:
        t = copy.deepcopy(ind.genome.create(8))
        specs.append(build_creature(t.getNodes(), ind.genome.moduleList)[0])
    ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), Morphology.from_specs(specs, 32).as_dict(), 2500,
                           n_threads=8, flags=oracle.FLAG_CONTINUOUS)["fitness"]
    fit = evaluate_population(inds)
    assert fit == ref.tolist()
    assert evaluate(inds[3]) == ref[3]


def test_merged_launch_equals_per_bucket_streams(need_gpu):
    """rem2d_worlds_step (all lane buckets in one grid) vs one rem2d_world_step per bucket: identical state."""
    import torch
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.env import BatchedModular2D
    specs = synthetic.lsystem_specs(range(96), mutate_odd=True)
    outs = []
    for merged in (True, False):
        env = BatchedModular2D(seed=4)
        env.merged_launch = merged
        env.reset_specs(specs)
        assert len(env.worlds) >= 3  # several lane buckets
        for n in (1, 3, 60, 120):
            env.step(n)
        torch.cuda.synchronize()
        outs.append(([w.bodies() for w, _ in env.worlds], env.fitness.cpu().numpy().copy(),
                     [w.view("cn0").cpu().numpy().copy() for w, _ in env.worlds], int(env.errors().max())))
        env.close()
    (b0, f0, c0, e0), (b1, f1, c1, e1) = outs
    assert e0 == 0 and e1 == 0
    assert all(np.array_equal(x, y) for x, y in zip(b0, b1))
    assert all(np.array_equal(x, y) for x, y in zip(c0, c1))
    assert np.array_equal(f0, f1)


def test_groups_step_one_call_and_graph_replay(need_gpu, oracle, rough_terrain):
    """rem2d_groups_step: the whole population's step groups in ONE ABI call (fork from the caller's stream, the steps of
    the groups queued round-robin, join), plain and replayed as a hipGraph (REM2D_STEP_GRAPH: captured on the first call
    of a given length, re-captured when a world's tiles change).  Fitness and steps equal the oracle's in every bit for
    one, three and four groups; an ABI misuse is refused."""
    import ctypes as C
    import torch
    from conftest import oracle_terrain
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.compiler import Morphology, lanes_for
    from gym_rem2d_amd.env import BatchedModular2D
    specs = synthetic.lsystem_specs(range(160), mutate_odd=True)
    ot = oracle_terrain(oracle, rough_terrain)
    ref = np.zeros(len(specs))
    groups = {}
    for e, sp in enumerate(specs):
        groups.setdefault(lanes_for(sp.n_bodies), []).append(e)
    T = 150
    for k in sorted(groups):
        m = Morphology.from_specs([specs[e] for e in groups[k]], k)
        ref[groups[k]] = oracle.batch_run(ot, m.as_dict(), T, n_threads=8, flags=oracle.FLAG_CONTINUOUS)["fitness"]
    # form 2 = the step train (the default), 1 = a launch sequence per step (what a hipGraph replays)
    for n_groups, graph, rebalance, form in ((1, False, 0, 2), (3, False, 0, 2), (4, False, 7, 2), (4, True, 0, 2), (1, False, 0, 1), (4, False, 0, 1),
                                             (4, True, 0, 1), (3, True, 0, 1), (4, True, 7, 1)):
        env = BatchedModular2D(seed=4, flags=_lib.FLAG_CONTINUOUS)
        env.step_groups = n_groups
        env.use_graph = graph
        env.rebalance_every = rebalance   # (under a graph the re-ordering launches of the captured call are replayed with it)
        env.reset_specs(specs)
        assert len(env.groups) == n_groups
        for w, _ in env.worlds:
            w.set_option("fuse_velpost", form)
        if not any(k in os.environ for k in ("REM2D_TILE_SHAPE", "REM2D_FUSE_VELPOST", "REM2D_PIPELINE", "REM2D_RETILE")):
            # rem2d_worlds_launch_info: 64-body tiles; as a step train (all steps of a call in one launch; with REM2D_STEP_GRAPH
            # the train is queued as it is -- one launch per call needs no graph) or velocity tiles + position iterations in one
            # launch per step
            assert env.launch_info() == (3, form)
        for n in (1, 24, 25, 25, 25, 25, 25):      # (25 five times: the graph of that length is captured once, replayed four times)
            env.step(n)
        torch.cuda.synchronize()
        assert np.array_equal(env.fitness.cpu().numpy(), ref), (n_groups, graph, rebalance, form)
        assert bool((env.steps == T).all()) and int(env.errors().max()) == 0
        env.close()
    L = _lib.lib()
    g = (_lib.StepGroup * 1)()
    g[0].worlds, g[0].n_worlds, g[0].stream = None, 0, None
    assert L.rem2d_groups_step(g, 1, 1, None, 0) == -1 and b"no worlds" in L.rem2d_last_error()
    assert L.rem2d_groups_step(g, _lib.MAX_STEP_GROUPS + 1, 1, None, 0) == -1
    assert L.rem2d_worlds_launch_info(None, 0, None, None) == -1


def test_episode_with_random_step_lengths_and_compactions(need_gpu, oracle):
    """An evaluate() episode driven at random (tools/fuzz_episode.py, a few rounds of it; 180 rounds:
    profiles/r04_fuzz_episode.txt): step calls of random lengths, compact() with random thresholds -- also while most creatures are
    alive and in mid-flight --, 1-4 step groups, device-made creature orders at random cadences, hipGraph replay or not.  The
    float64 fitness of every individual equals the oracle's."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_episode
    lines = []
    assert fuzz_episode.fuzz(5, 7, max_creatures=900, cap=400, report=lines.append) == 0, lines


def test_array_population_fitness_is_order_independent(need_gpu):
    """population.LSystemPopulation -> native compiler -> batched episodes: the fitness of an individual does not
    depend on where it sits in the population (bucket sorting / step groups / index gathering are transparent)."""
    import torch
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import run_episode
    from gym_rem2d_amd.population import LSystemPopulation
    rng = np.random.default_rng(11)
    pop = LSystemPopulation.random(600, rng, max_modules=15)
    env = BatchedModular2D()

    def evaluate(p):
        env.trees = env.robots = None
        env._upload(p.compile(2), len(p))
        return run_episode(env, max_steps=400).cpu().numpy()
    f1 = evaluate(pop)
    assert np.array_equal(f1, evaluate(pop))
    perm = rng.permutation(600)
    assert np.array_equal(evaluate(pop.select(perm)), f1[perm])
    assert (f1 > 0).sum() > 300 and len(np.unique(f1)) > 100
    env.close()


def test_skip_frozen_keeps_fitness_and_stops_stepping(need_gpu):
    """REM2D_FLAG_SKIP_FROZEN: wavefronts whose creatures all have a final fitness are no longer stepped; the
    fitness of every creature is the same as without the flag."""
    from gym_rem2d_amd import _lib
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import run_episode
    from gym_rem2d_amd.population import LSystemPopulation
    rng = np.random.default_rng(5)
    pop = LSystemPopulation.random(3000, rng, max_modules=15)
    out = []
    for flags in (_lib.FLAG_CONTINUOUS, _lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN):
        env = BatchedModular2D(flags=flags)
        env._upload(pop.compile(2), len(pop))
        fit = run_episode(env, max_steps=600).cpu().numpy()
        out.append((fit, env.steps.cpu().numpy(), env.frozen.cpu().numpy()))
        env.close()
    (f0, s0, z0), (f1, s1, z1) = out
    assert np.array_equal(f0, f1) and np.array_equal(z0, z1)
    assert s0.min() == s0.max()                       # without the flag everybody is stepped to the end
    assert s1.max() <= s0.max() and (s1 < s0).mean() > 0.3   # with it, most creatures stopped early


def test_compaction_of_the_survivors_keeps_every_fitness(need_gpu):
    """BatchedModular2D.compact (rem2d_world_adopt): between the chunks of an episode the creatures whose fitness is
    still open move into smaller worlds, state field by field; every fitness, frozen flag and error word is the one the
    uncompacted episode gives, for single- and multi-world populations, and the worlds do shrink."""
    import torch
    from gym_rem2d_amd import _lib
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import run_episode
    from gym_rem2d_amd.population import LSystemPopulation
    rng = np.random.default_rng(7)
    pop = LSystemPopulation.random(3000, rng, max_modules=15)
    flags = _lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN
    monkey_groups = {"REM2D_STEP_GROUPS": "3"}   # three worlds per lane bucket, as large populations have: they merge
    for batches in (pop.compile(2), [pop.compile(2)[-1]]):   # every lane bucket / the widest bucket alone (one world)
        n = sum(len(idx) for _, idx in batches)
        remap = {int(e): k for k, e in enumerate(np.concatenate([np.asarray(idx) for _, idx in batches]))}
        batches = [(m, [remap[int(e)] for e in idx]) for m, idx in batches]
        ref_env = BatchedModular2D(flags=flags)
        ref_env._upload(batches, n)
        ref = run_episode(ref_env, max_steps=600, compact=False)
        ref_frozen, ref_err = ref_env.frozen.clone(), ref_env.errors().clone()
        ref_env.close()
        env = BatchedModular2D(flags=flags)
        env.step_groups = int(monkey_groups["REM2D_STEP_GROUPS"])
        env._upload(batches, n)
        assert len(env.worlds) == 3 * len(batches)
        sizes = [sum(w.n_envs for w, _ in env.worlds)]
        done_steps, alive = 0, n
        while done_steps < 600 and alive > 0:
            env.step(50)
            done_steps += 50
            alive = env.compact(min_envs=32, max_alive=0.8)
            sizes.append(sum(w.n_envs for wi, (w, _) in enumerate(env.worlds) if wi not in env._inactive))
        assert torch.equal(env.fitness, ref)
        assert torch.equal(env.frozen, ref_frozen) and torch.equal(env.errors(), ref_err)
        assert sizes[-1] < 0.2 * sizes[0], sizes            # most creatures were dropped along the way
        rew, done = env.step(1) if alive else (env._reward, env._done)
        assert rew.shape[0] == n and done.shape[0] == n     # population order and size survive the compaction
        env.close()
    # the default path of run_episode compacts by itself and returns the same fitness
    env = BatchedModular2D(flags=flags)
    env._upload(pop.compile(2), len(pop))
    a = run_episode(env, max_steps=600)
    env.close()
    env = BatchedModular2D(flags=flags)
    env._upload(pop.compile(2), len(pop))
    b = run_episode(env, max_steps=600, compact=False)
    env.close()
    assert torch.equal(a, b)


def test_bench_json_contract(need_gpu):
    """bench.py prints exactly one JSON line with the fields the driver reads (tiny sizes here)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--envs", "1024", "--settle", "10"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["unit"] == "env-steps/s" and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # blocks of exactly `steps` env-steps, repeated until the timed region is >= --min-time (default 5 s); value = median
    cfg = d["config"]
    assert cfg["blocks"] == len(cfg["blocks_ms"]) >= 1 and (cfg["timed_region_s"] >= 5.0 or cfg["blocks"] == 400)
    assert d["creatures_total"] == cfg["creatures_total"] == 1024 and cfg["startup"]["to_first_block_s"] > 0
    import numpy as np
    assert abs(d["ms_per_step"] * d["steps"] - float(np.median(cfg["blocks_ms"]))) < 1e-2
    assert abs(d["value"] - cfg["creatures_total"] * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    assert d["metric"] == json.load(open(os.path.join(root, "BASELINE.json")))["metric"]
    # round 5: the line names the build it measured and the job's shape
    from gym_rem2d_amd import _lib
    assert d["build_id"] == _lib.source_id() and d["world_size"] == 1 and d["backend"] is None
    assert cfg["shard_cost_bodies"] and len(cfg["shard_cost_bodies"]) == 1 and cfg["shard_cost_bodies"][0] > 1024


def test_bench_gpus_flag_starts_the_ranks_itself(need_gpu):
    """`python bench.py --gpus 2` with no launcher around it starts two ranks (torch.distributed.run as a child) and
    reports n_gpus = 2; on a one-GPU box the ranks share the GPU and the fitness all-gather goes over gloo.  A rank
    count that contradicts the launcher's WORLD_SIZE is refused."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--envs", "512", "--settle", "5", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["metric"] == json.load(open(os.path.join(root, "BASELINE.json")))["metric"]   # the same string at every N
    assert d["config"]["envs_per_gpu"] == 512 and d["config"]["creatures_total"] == 1024 and d["config"]["solver_errors"] == 0
    # the line says how many ranks took part in the collective, over what, and how even the shards were
    assert d["world_size"] == 2 and d["backend"] in ("gloo", "nccl") and len(d["config"]["shard_cost_bodies"]) == 2
    assert min(d["config"]["shard_cost_bodies"]) > 512
    # strong scaling: the same population split over the ranks
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--envs", "512", "--settle", "5", "--no-cpu-baseline", "--scaling", "strong", "--min-time", "0"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["creatures_total"] == 512
    assert d["config"]["envs_per_gpu"] == 256 and d["config"]["blocks"] == 1 and d["config"]["solver_errors"] == 0
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True,
                         text=True, timeout=300, env=dict(env, WORLD_SIZE="2", RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def _plank_population(half_length=2.2):
    from gym_rem2d_amd import synthetic
    m = synthetic.chain_population(4, 2, "left")
    hx = m.arrays["hx"].reshape(4, m.lanes)
    hy = m.arrays["hy"].reshape(4, m.lanes)
    ang = m.arrays["angle"].reshape(4, m.lanes)
    hx[2, 0], hy[2, 0], ang[2, 0] = 0.1, half_length, np.float32(np.pi / 2)   # creature 2: a 4.4 m plank lying across ~9 edges
    return m


def test_creature_beyond_the_wide_build_gets_the_defined_penalty(need_gpu, oracle, flat_terrain):
    """Box2D has no contact cap (Modular2DEnv.py:634); the engine has two tiers (24 / 6, then 32 / 12 pair / solver slots per
    body).  A hand-built 7.6 m plank -- seven times the largest box the reference's classes can produce -- rests on more
    than 12 terrain edges: beyond both.  With on_error="penalty" (what the EA loops use) the episode completes, the plank
    gets evaluate.UNRESOLVED_FITNESS and is named in ``env.last_unresolved``; every other creature gets the oracle's
    fitness bit for bit, and the oracle -- asked how many touching manifolds its bodies held while the fitness was open --
    agrees on WHO is beyond the tiers, i.e. on every fitness under the same rule.  on_error="fallback" still raises."""
    import pytest
    import warnings
    from conftest import oracle_terrain
    from gym_rem2d_amd import _lib
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import UNRESOLVED_FITNESS, SolverOverflow, run_episode
    m = _plank_population(3.8)
    ref = oracle.batch_run_caps(oracle_terrain(oracle, flat_terrain), m.as_dict(), 150, n_threads=2, flags=oracle.FLAG_CONTINUOUS)
    pairs_cap, touch_cap = _lib.capacity(wide=True)
    beyond = (ref["caps"][:, 0] > pairs_cap) | (ref["caps"][:, 1] > touch_cap) | (ref["caps"][:, 2] > 0)
    assert beyond.tolist() == [False, False, True, False] and ref["caps"][2, 1] > touch_cap >= ref["caps"][[0, 1, 3], 1].max()
    want = np.where(beyond, UNRESOLVED_FITNESS, ref["fitness"])
    env = BatchedModular2D(flat=True)
    env.reset_morphology(m)
    with pytest.warns(UserWarning, match="UNRESOLVED_FITNESS"):
        fit = run_episode(env, max_steps=150, on_error="penalty")
    assert env.last_unresolved == [2] and env.last_overflow == [2]
    assert np.array_equal(fit.cpu().numpy(), want) and ref["fitness"][2] != UNRESOLVED_FITNESS
    env.reset_morphology(m)
    with pytest.raises(SolverOverflow):
        run_episode(env, max_steps=150)
    env.close()
    # the EA's own evaluator path (run_ea's default evaluate_batch, population.gpu_evaluator) runs in penalty mode: a
    # generation of ordinary individuals completes and reports nothing unresolved
    from gym_rem2d_amd import ea
    with warnings.catch_warnings():
        warnings.simplefilter("error")       # (no penalty warning for reference-legal creatures)
        pop, hist = ea.run_ea(ea.make_config(population_size=24, encoding="lsystem"), seed=5, n_generations=1, log=None)
    assert len(pop) == 24 and ea.run_ea.last_unresolved == [[], []]


def test_solver_overflow_falls_back_to_the_wide_build(need_gpu, oracle, flat_terrain):
    """A body resting on more than REM2D_SOLVER_SLOTS (6) terrain edges loses a manifold in the default build's solver
    (Modular2DEnv.py:634 would solve all of them: Box2D has no cap).  run_episode re-evaluates the flagged creature in
    the wide build (32 pair / 12 solver slots): every individual gets the oracle's fitness, bit for bit; the strict modes
    still raise / warn instead of handing a wrong fitness to selection."""
    import pytest
    import torch
    from conftest import oracle_terrain
    from gym_rem2d_amd import _lib
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import SolverOverflow, check_errors, run_episode
    from gym_rem2d_amd.world import BatchedWorld
    m = _plank_population()
    assert _lib.capacity() == (24, 6) and _lib.capacity(wide=True) == (32, 12)
    ref = oracle.batch_run(oracle_terrain(oracle, flat_terrain), m.as_dict(), 150, n_threads=2, flags=oracle.FLAG_CONTINUOUS)
    env = BatchedModular2D(flat=True)
    env.reset_morphology(m)
    fit = run_episode(env, max_steps=150)                                   # default: on_error="fallback"
    assert env.last_overflow == [2]
    assert np.array_equal(fit.cpu().numpy(), ref["fitness"])
    env.reset_morphology(m)
    with pytest.raises(SolverOverflow) as ei:
        run_episode(env, max_steps=150, on_error="raise")
    assert ei.value.indices == [2] and ei.value.codes[0] & _lib.ERR_SOLVER_OVERFLOW
    env.reset_morphology(m)
    with pytest.warns(UserWarning):
        fit = run_episode(env, max_steps=150, on_error="warn")
    assert fit.shape == (4,) and bool(check_errors(env, "ignore")[2]) and int(check_errors(env, "ignore").sum()) == 1
    env.close()
    # the wide build on its own: the plank's whole state == oracle (more than 6 touching manifolds on one body)
    w = BatchedWorld(m.n_envs, m.lanes, flags=_lib.FLAG_CONTINUOUS, wide=True)
    w.set_terrain(flat_terrain)
    w.reset(m)
    w.step(150)
    assert np.array_equal(w.bodies(), ref["bodies"]) and int(w.view("err").max()) == 0
    assert int((w.view("cinfo")[:, 2, 0] & 0xff).gt(0).sum()) > 6          # touching manifolds on the plank
    assert w.view("cedge").shape[0] == 32
    w.close()
    torch.cuda.synchronize()


def test_wide_build_is_the_same_engine(need_gpu, oracle, rough_terrain):
    """librem2d_wide.so is the same source with more slots: on populations that fit the default build it gives the same
    bits (== oracle), in one merged launch over several lane buckets and in the 256-lane tile shape."""
    from conftest import oracle_terrain
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.compiler import Morphology
    from gym_rem2d_amd.env import BatchedModular2D
    specs = synthetic.lsystem_specs(range(96))
    env = BatchedModular2D(flags=_lib.FLAG_CONTINUOUS, wide=True)
    env.reset_specs(specs)
    env.step(120)
    ot = oracle_terrain(oracle, rough_terrain)
    for w, idx in env.worlds:
        part = Morphology.from_specs([specs[e] for e in idx.cpu().tolist()], w.lanes)
        r = oracle.batch_run(ot, part.as_dict(), 120, n_threads=8, flags=oracle.FLAG_CONTINUOUS)
        assert w.wide and np.array_equal(w.bodies(), r["bodies"])
        assert np.array_equal(w.view("fitness").cpu().numpy(), r["fitness"])
    assert int(env.errors().max()) == 0
    env.close()


def test_state_dump_for_external_viewers(need_gpu, tmp_path):
    """statedump.record_episode: JSON lines (header with terrain + morphology, then frames with poses / wall of death /
    reward) and the draw list Modular2D.render would paint (Modular2DEnv.py:655-738)."""
    import json
    from gym_rem2d_amd import synthetic, statedump
    from gym_rem2d_amd.env import BatchedModular2D
    specs = synthetic.lsystem_specs(range(6))
    env = BatchedModular2D()
    env.reset_specs(specs)
    path = statedump.record_episode(env, str(tmp_path / "run.jsonl"), steps=40, creatures=(0, 4), every=10)
    lines = open(path).read().splitlines()
    head, frames = json.loads(lines[0]), [json.loads(l) for l in lines[1:]]
    assert head["kind"] == "rem2d_state_dump" and len(head["terrain"]["x"]) == 200 and len(frames) == 5
    assert [c["index"] for c in head["creatures"]] == [0, 4]
    assert [len(c["bodies"]) for c in head["creatures"]] == [specs[0].n_bodies, specs[4].n_bodies]
    assert frames[0]["step"] == 0 and frames[-1]["step"] == 40
    assert frames[-1]["creatures"][0]["wall_of_death"] == pytest.approx(40 * 0.04)
    y0, y1 = frames[0]["creatures"][0]["pose"][0][1], frames[-1]["creatures"][0]["pose"][0][1]
    assert y1 < y0                                        # it fell from the spawn height
    prims = statedump.frame_to_draw_list(head, frames[-1])
    assert len(prims) == 2 and sum(p[0] in ("polygon", "circle") for p in prims[0]) == specs[0].n_bodies
    env.close()


def test_step_returns_population_order_reward_and_done(need_gpu):
    """A mixed population lives in one world per lane count (and several step groups); the kernels write reward / done
    straight into population-order arrays (rem2d_world_set_outputs), so what step() returns (Modular2DEnv.py:642-653)
    equals the per-world fields gathered by hand -- in every step, with no gather kernels on the way."""
    import torch
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.env import BatchedModular2D
    specs = synthetic.lsystem_specs(range(300))
    env = BatchedModular2D()
    env.reset_specs(specs)
    assert len(env.worlds) >= 3
    for n in (1, 1, 7, 40, 130):
        reward, done = env.step(n)
        torch.cuda.synchronize()
        ref_r = torch.zeros_like(reward)
        ref_d = torch.zeros_like(done)
        for w, idx in env.worlds:
            ref_r[idx] = w.view("reward")
            ref_d[idx] = w.view("done") != 0
        assert reward.dtype == torch.float32 and done.dtype == torch.bool and reward.shape == (300,)
        assert torch.equal(reward, ref_r) and torch.equal(done, ref_d)
    assert bool(done.any()) and not bool(done.all())      # the wall of death has caught some creatures by step 179
    env.close()


def test_render_rgb_array(need_gpu):
    """Modular2D.render(mode='rgb_array') (Modular2DEnv.py:655-738 without pyglet): an image of the current state."""
    import random
    from gym_rem2d_amd import get_module_list
    from gym_rem2d_amd.encodings import LSystem
    from gym_rem2d_amd.env import Modular2D
    random.seed(3)
    ml = get_module_list()
    g = LSystem(ml)
    env = Modular2D()
    env.seed(4)
    env.reset(tree=g.create(8), module_list=ml)
    for _ in range(30):
        env.step(None)
    img = env.render(mode="rgb_array")
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3 and img.shape[0] > 100 and img.shape[1] > 100
    assert len(np.unique(img.reshape(-1, 3), axis=0)) > 3          # terrain, module colours, background
    with pytest.raises(NotImplementedError):
        env.render(mode="human")
    env.close()


def test_bench_population_exactly_as_benched_equals_the_oracle(need_gpu, oracle, monkeypatch, tmp_path):
    """The EXACT population bench.py's default line is quoted on -- config 3: 65 536 L-system creatures, seeds 0..65535,
    built by bench.build_population, uploaded by bench.make_env (four lane buckets x four step groups, merged launches,
    continuous physics), stepped like the bench steps it (settle 60 + 60 steps, 25 per ABI call) -- with a 2 % sample of
    every world re-run by the oracle from reset: every pose, velocity, sleep timer and awake flag identical, no error bit
    anywhere in the population, no NaN."""
    import tempfile
    import torch
    import bench
    from gym_rem2d_amd import _lib, make_terrain
    monkeypatch.setattr(tempfile, "gettempdir", lambda: str(tmp_path))
    monkeypatch.setenv("REM2D_BENCH_NO_FORK", "1")            # (this process has initialised the GPU: no fork pool)
    morphs, desc = bench.finish_population(bench.build_population("lsystem", 65536, 0))
    assert "65536 random L-System creatures (seeds 0..65535" in desc and [m.lanes for m in morphs] == [2, 4, 8, 16]
    dev = torch.device("cuda", 0)
    env = bench.make_env(morphs, dev, False, True, False)
    assert len(env.groups) == 1 and len(env.worlds) == 4 and env.launch_info() == (3, 2)   # (the step train: one group)
    T = 120
    bench.stepper(env, 25)(T)
    torch.cuda.synchronize()
    assert int(env.errors().max()) == 0
    ot = oracle_terrain(oracle, make_terrain(4, flat=True))
    checked = 0
    for (w, _), part in zip(env.worlds, env._world_morph):
        got = w.bodies()
        assert np.isfinite(got).all()
        pick = np.arange(0, part.n_envs, 50)
        ref = oracle.batch_run(ot, part.take(pick).as_dict(), T, n_threads=16, flags=oracle.FLAG_CONTINUOUS)
        assert np.array_equal(got[pick], ref["bodies"]), "world of %d-lane creatures differs from the oracle" % w.lanes
        assert np.array_equal(w.view("fitness").cpu().numpy()[pick], ref["fitness"])
        checked += len(pick)
    assert checked >= 0.02 * 65536
    env.close()


def test_rebalance_by_current_cost_keeps_population_order_results(need_gpu, oracle, rough_terrain):
    """Mixed populations re-order their creatures by current cost every REBALANCE_EVERY env-steps (launch option `rebalance`:
    the creatures that used every position iteration move to the front of their world's order, on the device); the host-side
    BatchedModular2D.rebalance() installs the same kind of order through rem2d_world_set_order.  What step() returns stays
    in POPULATION order and equals the oracle bit for bit; the host-made orders are permutations with the slow creatures
    first."""
    import torch
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.env import BatchedModular2D
    specs = synthetic.lsystem_specs(range(6000, 6000 + 4608), mutate_odd=True)
    env = BatchedModular2D(flags=_lib.FLAG_CONTINUOUS)
    env.reset_specs(specs)
    assert env._rebalance_steps == env.REBALANCE_EVERY and not (env._world_flags & _lib.FLAG_RETILE)
    assert all(w.get_option("rebalance") == env.REBALANCE_EVERY for w, _ in env.worlds)
    for k in range(6):
        reward, done = env.step(25)
        if k == 3:
            env.rebalance()                  # (the host-side form on top, in mid-run)
    torch.cuda.synchronize()
    orders = [w._order[:w.n_envs].cpu().numpy() for w, _ in env.worlds if getattr(w, "_order", None) is not None]
    assert orders and all(np.array_equal(np.sort(o), np.arange(len(o))) for o in orders)
    assert any((o != np.arange(len(o))).any() for o in orders)        # some creature did move
    ot = oracle_terrain(oracle, rough_terrain)
    fit = env.fitness.cpu().numpy()
    for (w, idx), part in zip(env.worlds, env._world_morph):
        r = oracle.batch_run(ot, part.as_dict(), 150, n_threads=8, flags=oracle.FLAG_CONTINUOUS)
        pop = idx.cpu().numpy()
        assert np.array_equal(w.bodies(), r["bodies"])
        assert np.array_equal(fit[pop], r["fitness"])
        assert np.array_equal(reward.cpu().numpy()[pop], r["reward"].astype(np.float32))
    assert int(env.errors().max()) == 0
    env.close()
    uni = BatchedModular2D(flat=True, flags=_lib.FLAG_CONTINUOUS)      # fixed-morphology populations have nothing to re-order
    uni.reset_morphology(synthetic.chain_population(8192, 4, "left"))
    assert uni._rebalance_steps == 0 and uni.worlds[0][0].get_option("rebalance") == 0
    uni.close()


def test_five_joints_on_one_body_fall_back_to_the_wide_build(need_gpu, oracle, flat_terrain):
    """A module of the reference has three connection sites and the joint to its parent (simple_module.py:21-25,
    circular_module.py:23-27): at most four joints on a body, schedule period <= 4 -- what the default build unrolls its tick loop
    for (V4_PHASES, round 6).  A hand-built hub with FIVE children (period 5) is outside that domain: the default build flags it
    REM2D_ERR_SOLVER_OVERFLOW like a body beyond its contact slots, run_episode re-evaluates it in the wide build (which keeps five
    phases), and every creature -- the star and the legal period-4 creatures beside it -- gets the oracle's fitness bit for bit."""
    from gym_rem2d_amd import _lib
    from gym_rem2d_amd.compiler import CreatureBuilder, CreatureSpec, Morphology
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import run_episode

    def star(n_children):
        b = CreatureBuilder()
        hub = b.add_box(0.25, 0.25, 5.0, 7.0, 0.0)
        for k in range(n_children):
            ang = 2.0 * np.pi * k / n_children
            child = b.add_box(0.1, 0.3, 5.0 + 0.55 * np.cos(ang), 7.0 + 0.55 * np.sin(ang), ang - np.pi / 2)
            b.add_revolute(hub, child, (0.25 * np.cos(ang), 0.25 * np.sin(ang)), (0.0, -0.3), 50.0)
        return CreatureSpec(b, list(range(n_children + 1)))
    specs = [star(4), star(5), star(3), star(4)] + [star(4 - k % 2) for k in range(8)]
    assert [s.period for s in specs[:4]] == [4, 5, 3, 4] and max(s.period for s in specs[4:]) == 4
    m = Morphology.from_specs(specs, 8)
    ref = oracle.batch_run(oracle_terrain(oracle, flat_terrain), m.as_dict(), 300, n_threads=4, flags=oracle.FLAG_CONTINUOUS)
    env = BatchedModular2D(flat=True)
    env.reset_morphology(m)
    env.step(5)
    err = env.errors().cpu().numpy()
    # (a velocity tile that cannot schedule one of its creatures leaves joints of ANY of them unsolved: the eight creatures of the
    # star's 64-lane tile carry the flag, the second tile's legal creatures do not)
    assert err.tolist() == [_lib.ERR_SOLVER_OVERFLOW] * 8 + [0] * 4
    env.reset_morphology(m)
    fit = run_episode(env, max_steps=300)
    assert env.last_overflow == list(range(8)) and env.last_unresolved == []
    assert np.array_equal(fit.cpu().numpy(), ref["fitness"])
    env.close()
