"""Population sharding + fitness all-gather, world_size 2 over gloo on CPU.  The local evaluator in
this test is the oracle (tests may use it as the checker); on the GPU box it is the HIP stepper."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gym_rem2d_amd.evaluate import shard_balanced, shard_costs, shard_range


def test_shard_range_covers_everything():
    for n in (1, 7, 64, 65536, 1000003):
        for W in (1, 2, 3, 8):
            blocks = [shard_range(n, r, W) for r in range(W)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            assert max(hi - lo for lo, hi in blocks) == -(-n // W)


def test_shard_balanced_deals_every_individual_once_and_evens_the_cost():
    """The static deal that replaces pool.map's dynamic balancing (REM2D_main.py:256-262): descending cost, snake-wise."""
    rng = np.random.default_rng(0)
    for n in (1, 7, 64, 1001, 65536):
        cost = rng.integers(2, 17, n)               # bodies per creature
        for W in (1, 2, 3, 8):
            idx = shard_balanced(cost, W)
            assert idx.shape == (W, -(-n // W))
            flat = idx[idx >= 0]
            assert np.array_equal(np.sort(flat), np.arange(n))          # everyone exactly once
            assert (idx >= 0).sum(axis=1).max() - (idx >= 0).sum(axis=1).min() <= 1
            c = shard_costs(cost, idx)
            if n >= 1001:
                assert c.max() - c.min() <= 16 and c.max() / c.min() < 1.01   # within one creature's cost of each other
    # a population sorted by cost is the worst case of a contiguous cut and no case at all for the deal
    cost = np.sort(rng.integers(2, 17, 4096))[::-1]
    contiguous = np.array([cost[lo:hi].sum() for lo, hi in (shard_range(4096, r, 8) for r in range(8))])
    dealt = shard_costs(cost, shard_balanced(cost, 8))
    assert contiguous.max() / contiguous.min() > 3 and dealt.max() / dealt.min() < 1.01
    # ties keep the population order: the deal is a pure function of the key (every rank computes the same one)
    assert np.array_equal(shard_balanced(np.ones(6), 2), [[0, 3, 4], [1, 2, 5]])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, steps, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_rem2d_amd import make_terrain, synthetic
    from gym_rem2d_amd.compiler import Morphology
    from gym_rem2d_amd.evaluate import evaluate_population_sharded
    from oracle import oracle as O
    terrain = make_terrain(4)
    xs, ys, _ = terrain.f32()
    ot = O.Terrain(xs, ys)
    specs = synthetic.lsystem_specs(range(n_total))
    morph = Morphology.from_specs(specs, 32)

    def local_eval(lo, hi):
        if hi <= lo:
            return torch.zeros(0)
        r = O.batch_run(ot, morph.take(np.arange(lo, hi)).as_dict(), steps, n_threads=1)
        return torch.from_numpy(r["fitness"])

    fit = evaluate_population_sharded(n_total, local_eval)
    np.save(os.path.join(out_dir, "fit%d.npy" % rank), fit.numpy())
    # the same job with cost-balanced shards: the ranks' individuals are dealt by body count, local_eval gets index arrays,
    # the fitness comes back in population order through the same single all-gather

    def local_eval_idx(idx):
        if len(idx) == 0:
            return torch.zeros(0)
        r = O.batch_run(ot, morph.take(np.asarray(idx)).as_dict(), steps, n_threads=1)
        return torch.from_numpy(r["fitness"])
    fit2 = evaluate_population_sharded(n_total, local_eval_idx, cost=morph.n_bodies)
    np.save(os.path.join(out_dir, "fitb%d.npy" % rank), fit2.numpy())
    np.save(os.path.join(out_dir, "cost%d.npy" % rank), evaluate_population_sharded.last_shard_cost)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_evaluation_gloo(tmp_path, oracle):
    n_total, steps, world = 11, 60, 2  # odd count: the last rank's block is shorter (padding path)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, steps, str(tmp_path)), nprocs=world, join=True)
    from gym_rem2d_amd import make_terrain, synthetic
    from gym_rem2d_amd.compiler import Morphology
    terrain = make_terrain(4)
    xs, ys, _ = terrain.f32()
    ot = oracle.Terrain(xs, ys)
    morph = Morphology.from_specs(synthetic.lsystem_specs(range(n_total)), 32)
    ref = oracle.batch_run(ot, morph.as_dict(), steps, n_threads=2)["fitness"]
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "fit%d.npy" % r))
        # float64 end to end: the gathered values are the evaluators' doubles, bit for bit (REM2D_main.py:372-375)
        assert got.dtype == np.float64 and got.shape == (n_total,) and np.array_equal(got, ref)
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "fitb%d.npy" % r)), ref)      # dealt shards: same values
        cost = np.load(os.path.join(str(tmp_path), "cost%d.npy" % r))
        assert cost.shape == (world,) and cost.sum() == morph.n_bodies.sum() and abs(cost[0] - cost[1]) <= morph.n_bodies.max()


def _ea_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    g.build()
    from gym_rem2d_amd.population import LSystemPopulation, run_generations, sharded_evaluator

    def local_eval(block):   # CPU stand-in for the batched episode: any deterministic function of the phenotype
        out = np.zeros(len(block), dtype=np.float32)
        for m, idx in block.compile(1):
            x = m.arrays["x"].reshape(m.n_envs, m.lanes).astype(np.float64)
            out[np.asarray(idx)] = (x.sum(axis=1) + m.n_bodies).astype(np.float32)
        return out

    rng = np.random.default_rng(7)          # the same seed on every rank: replicated selection / mutation
    pop = LSystemPopulation.random(301, rng, max_modules=15)
    ev = sharded_evaluator(local_eval)      # (balance=True: shards dealt by body count)
    pop, fit, hist = run_generations(pop, 3, ev, rng, 0.2, 0.2, 0.2)
    np.save(os.path.join(out_dir, "ea_cost%d.npy" % rank), ev.last_shard_cost)
    np.save(os.path.join(out_dir, "ea_fit%d.npy" % rank), fit)
    np.save(os.path.join(out_dir, "ea_angle%d.npy" % rank), pop.a["mod_angle"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_array_ea_gloo(tmp_path):
    """Array EA on 2 ranks: replicated variation + sharded evaluation (cost-balanced shards: individuals dealt by body count)
    + one fitness all-gather per generation gives every rank the same population and the same fitness as a single process."""
    world, port = 2, _free_port()
    mp.spawn(_ea_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from gym_rem2d_amd.population import LSystemPopulation, run_generations

    def evaluate(p):
        out = np.zeros(len(p), dtype=np.float32)
        for m, idx in p.compile(1):
            x = m.arrays["x"].reshape(m.n_envs, m.lanes).astype(np.float64)
            out[np.asarray(idx)] = (x.sum(axis=1) + m.n_bodies).astype(np.float32)
        return out.astype(np.float64)
    rng = np.random.default_rng(7)
    pop = LSystemPopulation.random(301, rng, max_modules=15)
    pop, fit, hist = run_generations(pop, 3, evaluate, rng, 0.2, 0.2, 0.2)
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "ea_fit%d.npy" % r)), fit)
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "ea_angle%d.npy" % r)), pop.a["mod_angle"])
        # both ranks' predicted cost (bodies to step) within 5 % of each other, and together the whole population
        cost = np.load(os.path.join(str(tmp_path), "ea_cost%d.npy" % r))
        assert abs(cost[0] - cost[1]) <= 0.05 * cost.mean() and cost.sum() == pop.body_counts(1).sum()
    assert hist[-1][3] >= hist[0][3]   # selection pushes the mean up


def _overflow_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_rem2d_amd.evaluate import SolverOverflow, evaluate_population_sharded
    from gym_rem2d_amd.population import LSystemPopulation, sharded_evaluator
    n_total = 9

    def local_eval(lo, hi):   # rank 1 holds a creature (population index 7) without a valid fitness
        fit = torch.arange(lo, hi, dtype=torch.float64)
        bad = torch.zeros(hi - lo, dtype=torch.bool)
        if lo <= 7 < hi:
            bad[7 - lo] = True
        return fit, bad

    verdict = "completed"
    try:
        evaluate_population_sharded(n_total, local_eval, on_error="raise")
    except SolverOverflow as e:      # raised AFTER the collective, on every rank, with the global index
        verdict = "overflow %s" % e.indices
    # the library default raises too; an EA loop opts in to the defined penalty: on every rank alike, the job goes on
    try:
        evaluate_population_sharded(n_total, local_eval)
        verdict += " (default completed)"
    except SolverOverflow:
        pass
    import warnings
    from gym_rem2d_amd.evaluate import UNRESOLVED_FITNESS
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fit = evaluate_population_sharded(n_total, local_eval, on_error="penalty")
    want = torch.arange(n_total, dtype=torch.float64)
    want[7] = UNRESOLVED_FITNESS
    assert torch.equal(fit, want) and evaluate_population_sharded.last_unresolved == [7]
    # the array EA's evaluator: same protocol
    rng = np.random.default_rng(3)
    pop = LSystemPopulation.random(10, rng, max_modules=15)

    def block_eval(block):
        bad = np.zeros(len(block), dtype=bool)
        if rank == 1:
            bad[0] = True
        return np.ones(len(block)), bad
    try:
        sharded_evaluator(block_eval, on_error="raise", balance=False)(pop)   # (contiguous blocks: rank 1 starts at index 5)
        verdict += " | completed"
    except SolverOverflow as e:
        verdict += " | overflow %s" % e.indices
    ev = sharded_evaluator(block_eval, on_error="penalty", balance=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f2 = ev(pop)
    assert ev.last_unresolved == [5] and f2[5] == UNRESOLVED_FITNESS and (np.delete(f2, 5) == 1.0).all()
    # with cost-balanced shards (the default) rank 1's first individual is whoever the deal gives it: same protocol
    from gym_rem2d_amd.evaluate import shard_balanced
    evb = sharded_evaluator(block_eval, on_error="penalty")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f3 = evb(pop)
    who = int(shard_balanced(pop.body_counts(1), 2)[1][0])
    assert evb.last_unresolved == [who] and f3[who] == UNRESOLVED_FITNESS and (np.delete(f3, who) == 1.0).all()
    assert ev.unresolved_log == [[5]]   # (one entry per call: a long run keeps every generation's list)
    # and a clean job still returns the gathered fitness
    fit = evaluate_population_sharded(n_total, lambda lo, hi: (torch.arange(lo, hi, dtype=torch.float64),
                                                               torch.zeros(hi - lo, dtype=torch.bool)))
    assert torch.equal(fit, torch.arange(n_total, dtype=torch.float64))
    with open(os.path.join(out_dir, "verdict%d.txt" % rank), "w") as f:
        f.write(verdict)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_job_with_an_overflowing_creature_completes_gloo(tmp_path):
    """A rank whose shard holds a creature without a valid fitness (it overflowed even the wide build) must not raise
    before the job's collective -- the other ranks would wait in the all-gather for ever.  The mask rides in the fitness
    all-gather; afterwards every rank holds the same verdict: with on_error="raise" (the library default: the reference has no
    contact cap and would have produced a fitness) EVERY rank raises SolverOverflow, naming the population index; with
    on_error="penalty" (what an EA loop opts into, so that one out-of-domain creature does not abort a generation) the creature
    gets evaluate.UNRESOLVED_FITNESS on every rank alike and the job goes on."""
    world, port = 2, _free_port()
    mp.spawn(_overflow_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "verdict%d.txt" % r)).read() == "overflow [7] | overflow [5]"
