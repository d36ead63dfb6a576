"""Array populations of the direct and the network encoding (batched genotype -> phenotype and the EA
loop without per-individual Python objects; REM2D_main.py:280-298 mutates every offspring of every generation).  CPU: the node
tables keep the reference's tree invariants, the vectorised operators reproduce the object genomes' distributions
(Encodings/Direct_Encoding.py:82-139, Network_Encoding.py:142-152, simple_module.py:70-85, m_controller.py:50-58), and
compile() == the object path's expression.  GPU: a generation with the batched episode as evaluator == the same generation
with the oracle as evaluator."""
import copy
import random

import numpy as np
import pytest

from gym_rem2d_amd import ea
from gym_rem2d_amd.population import DirectPopulation, LSystemPopulation, NetworkPopulation, run_generations


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()


def _same_batches(b1, b2):
    assert [(m.lanes, m.n_envs) for m, _ in b1] == [(m.lanes, m.n_envs) for m, _ in b2]
    for (m1, i1), (m2, i2) in zip(b1, b2):
        assert list(i1) == list(i2) and np.array_equal(m1.n_bodies, m2.n_bodies)
        for k in m1.arrays:
            assert np.array_equal(m1.arrays[k], m2.arrays[k]), k


def test_direct_population_tables_are_the_object_genomes_trees(built):
    """from_genomes: DirectEncoding objects -> node tables; compile() gives the very morphologies of the object path."""
    from gym_rem2d_amd.encode import encode_trees_native
    random.seed(11)
    inds = [ea.Individual.random(encoding="direct") for _ in range(60)]
    for ind in inds[::2]:
        ind.mutate(0.3, 0.3, 0.3)
    pop = DirectPopulation.from_genomes([i.genome for i in inds])
    assert pop.check() and len(pop) == 60
    assert [int(c) for c in pop.a["node_count"]] == [len(i.genome.create(8).getNodes()) for i in inds]
    _same_batches(pop.compile(2), encode_trees_native(inds, 8, n_threads=2))
    assert np.array_equal(pop.body_counts(2), np.concatenate([m.n_bodies for m, _ in pop.compile(2)])[
        np.argsort(np.concatenate([np.asarray(i) for _, i in pop.compile(2)]))])
    # select() clones: mutating the clone leaves the parent alone
    kid = pop.select([3, 3, 7])
    before = {k: v.copy() for k, v in pop.a.items()}
    kid.mutate(0.5, 0.5, 0.5, np.random.default_rng(0))
    assert all(np.array_equal(before[k], pop.a[k]) for k in before) and kid.check()


def test_direct_population_mutates_like_the_object_genome(built):
    """Same distributions as DirectEncoding.mutate, event for event (removal with the skipped sibling, the root's children that
    are only not descended into, growth with the skipped site and the recounted n_modules, visited nodes only mutate and get
    clamped): tree sizes, depths, shapes and parameters of 3 000 individuals after the five initial rounds and after 25 more."""
    from gym_rem2d_amd.encodings.direct import DirectEncoding
    from gym_rem2d_amd.modules import get_module_list
    random.seed(1)
    N = 3000
    objs = [DirectEncoding(get_module_list()) for _ in range(N)]
    rng = np.random.default_rng(0)
    pops = {"native": DirectPopulation.random(N, rng), "numpy": DirectPopulation.random(N, rng, numpy_only=True)}

    def object_stats():
        sizes, widths, amps, boxes = [], [], [], []
        for g in objs:
            nodes = g.tree.getNodes()
            sizes.append(len(nodes))
            for nd in nodes:
                boxes.append(nd.module_.type == "SIMPLE")
                amps.append(nd.controller.amplitude)
                if nd.module_.type == "SIMPLE":
                    widths.append(nd.module_.width)
        return np.array(sizes), np.array(widths), np.array(amps), np.mean(boxes)

    def array_stats(pop):
        live = pop.a["shape"] > 0
        return (pop.a["node_count"], pop.a["width"][pop.a["shape"] == 1], pop.a["ctl_amp"][live],
                (pop.a["shape"] == 1).sum() / live.sum())

    for rounds in (0, 25):
        for _ in range(rounds):
            for g in objs:
                g.mutate(0.1, 0.1, 0.1)
            pops["native"].mutate(0.1, 0.1, 0.1, rng, n_threads=2)       # rem2d_mutate_trees (the product path)
            pops["numpy"].mutate_numpy(0.1, 0.1, 0.1, rng)               # the same operator, vectorised numpy
        s1, w1, a1, b1 = object_stats()
        for name, pop in pops.items():
            assert pop.check()
            s2, w2, a2, b2 = array_stats(pop)
            assert abs(s1.mean() - s2.mean()) < 0.12 and abs(s1.std() - s2.std()) < 0.12, (name, rounds, s1.mean(), s2.mean())
            assert abs(w1.mean() - w2.mean()) < 0.01 and abs(w1.std() - w2.std()) < 0.01, (name, rounds, w1.mean(), w2.mean())
            assert abs((w1 == 0.2).mean() - (w2 == 0.2).mean()) < 0.01   # never-visited nodes keep the un-clamped default
            assert abs(a1.mean() - a2.mean()) < 0.02 and abs(b1 - b2) < 0.03
            h1, h2 = np.bincount(s1, minlength=21) / N, np.bincount(s2, minlength=21) / N
            assert np.abs(h1 - h2).max() < 0.03, (name, rounds, h1, h2)
            assert pop.a["node_count"].max() <= 20 and (pop.depths().max() <= 8)
    # the native mutation does not depend on the thread count (one generator per individual)
    p1 = DirectPopulation.random(500, np.random.default_rng(9))
    p2 = p1.select(np.arange(500))
    p1.mutate(0.3, 0.3, 0.3, np.random.default_rng(1), n_threads=1)
    p2.mutate(0.3, 0.3, 0.3, np.random.default_rng(1), n_threads=5)
    assert all(np.array_equal(p1.a[k], p2.a[k]) for k in p1.a)


def test_network_population_compiles_and_mutates_like_the_object_genome(built):
    from gym_rem2d_amd.encode import encode_network_native
    random.seed(4)
    inds = [ea.Individual.random(encoding="cppn") for _ in range(48)]
    for ind in inds[::3]:
        ind.mutate(0.3, 0.3, 0.3)
    depth = inds[0].tree_depth
    pop = NetworkPopulation.from_genomes([i.genome for i in inds], tree_depth=depth)
    _same_batches(pop.compile(2), encode_network_native(inds, depth, n_threads=2))
    order = np.argsort(np.concatenate([np.asarray(i) for _, i in pop.compile(2)]))
    assert np.array_equal(pop.body_counts(2), np.concatenate([m.n_bodies for m, _ in pop.compile(2)])[order])
    # FeedForwardCPPN.mutate: every weight with probability 0.2 += gauss(0, 0.3); prototypes like the other populations
    rng = np.random.default_rng(2)
    big = NetworkPopulation.random(4000, rng)
    w1, w2 = big.a["w1"].copy(), big.a["w2"].copy()
    assert abs(w1.std() - 1.5) < 0.02 and abs(w2.std() - 1.0) < 0.02 and set(np.unique(big.a["a1"])) == {0, 1, 2, 3}
    big.mutate(0.1, 0.1, 0.1, rng)
    d = np.concatenate([(big.a["w1"] - w1).ravel(), (big.a["w2"] - w2).ravel()])
    assert abs((d != 0).mean() - 0.2) < 0.005 and abs(d[d != 0].std() - 0.3) < 0.01
    kid = big.select(np.arange(10))
    kid.a["w1"][:] = 0
    assert (big.a["w1"][:10] != 0).any()                                 # select() clones


def test_run_generations_takes_every_array_population(built):
    """The generational loop on arrays (population.run_generations) with a stub evaluator: any of the three populations."""
    for make in (lambda r: LSystemPopulation.random(64, r, max_modules=15), lambda r: DirectPopulation.random(64, r),
                 lambda r: NetworkPopulation.random(64, r)):
        rng = np.random.default_rng(5)
        pop = make(rng)
        pop2, fit, hist = run_generations(pop, 3, lambda p: p.body_counts(1).astype(np.float64), rng, 0.2, 0.2, 0.2)
        assert len(pop2) == 64 and len(hist) == 3 and type(pop2) is type(pop) and hist[-1][3] >= hist[0][3] - 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("encoding", ["direct", "network"])
def test_array_generation_on_gpu_equals_generation_with_the_oracle(encoding, oracle):
    """(f1 / f2) tournament -> clone -> mutate -> native expression -> batched episodes for the array populations of the direct
    and the network encoding: the GPU evaluator and the oracle as evaluator give the same fitness, bit for bit, hence the
    same offspring in every generation."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gym_rem2d_amd import make_terrain
    from gym_rem2d_amd.population import gpu_evaluator
    steps = 400
    terrain = make_terrain(4)
    xs, ys, _ = terrain.f32()
    ot = oracle.Terrain(xs, ys, None, terrain.friction)

    def oracle_eval(pop):
        out = np.zeros(len(pop))
        for m, idx in pop.compile(2):
            out[np.asarray(idx)] = oracle.batch_run(ot, m.as_dict(), steps, n_threads=8, flags=oracle.FLAG_CONTINUOUS)["fitness"]
        return out

    def make(rng):
        return DirectPopulation.random(96, rng) if encoding == "direct" else NetworkPopulation.random(96, rng)
    res = []
    for ev in (gpu_evaluator(max_steps=steps), oracle_eval):
        rng = np.random.default_rng(17)
        res.append(run_generations(make(rng), 2, ev, rng, 0.2, 0.2, 0.2))
    (p1, f1, h1), (p2, f2, h2) = res
    assert np.array_equal(f1, f2) and h1 == h2 and len(set(f1.tolist())) > 10
    for k in p1.a:
        assert np.array_equal(p1.a[k], p2.a[k]), k


def test_count_only_compilers_and_error_paths(built):
    """out == NULL turns the three native compilers into body counters (the cost key of evaluate.shard_balanced); a node table that
    is not a tree in Tree.getNodes() order is refused by rem2d_mutate_trees with an error code, not a crash."""
    import ctypes as C
    from gym_rem2d_amd import _lib
    rng = np.random.default_rng(3)
    ls = LSystemPopulation.random(300, rng, max_modules=15)
    order = np.argsort(np.concatenate([np.asarray(i) for _, i in ls.compile(2)]))
    assert np.array_equal(ls.body_counts(2), np.concatenate([m.n_bodies for m, _ in ls.compile(2)])[order])
    pop = DirectPopulation.random(50, rng)
    good = {k: v.copy() for k, v in pop.a.items()}
    pop.a["parent"][7, 1] = 5                       # a parent BEHIND its child: not pre-order
    with pytest.raises(_lib.Rem2dError, match="Tree.getNodes"):
        pop.mutate(0.1, 0.1, 0.1, rng)
    pop.a = {k: v.copy() for k, v in good.items()}
    pop.a["node_count"][3] = 0
    with pytest.raises(_lib.Rem2dError):
        pop.mutate(0.1, 0.1, 0.1, rng)
    assert _lib.lib().rem2d_mutate_trees(None, 0.1, 0.1, 0.1, 1, 1) == -1
    # morph_rate 0: no structural change, only (some) parameters move; rate 0 as well: nothing moves at all
    pop.a = {k: v.copy() for k, v in good.items()}
    pop.mutate(0.0, 0.0, 0.1, rng)
    for k in ("node_count", "parent", "site", "shape", "ctl_amp", "ctl_phase", "ctl_freq"):
        assert np.array_equal(pop.a[k], good[k]), k
    live = good["shape"] > 0                        # (limitWH / minMax of the visited nodes may still clamp: width 0.2 -> 0.5)
    assert (pop.a["width"][live & (good["shape"] == 1)] >= 0.5).all()


def test_shard_balanced_edge_cases():
    from gym_rem2d_amd.evaluate import shard_balanced, shard_costs
    assert shard_balanced(np.zeros(0), 4).shape == (4, 0)
    idx = shard_balanced(np.array([5, 9, 2]), 8)                       # more ranks than individuals
    assert idx.shape == (8, 1) and sorted(idx[idx >= 0].tolist()) == [0, 1, 2] and (idx >= 0).sum() == 3
    assert shard_costs(np.array([5, 9, 2]), idx).sum() == 16
    assert np.array_equal(shard_balanced(np.array([1, 8, 3, 8]), 1), [[1, 3, 2, 0]])   # ties keep population order


def _sharded_worker(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    g.build()
    from gym_rem2d_amd.population import DirectPopulation, NetworkPopulation, run_generations, sharded_evaluator
    for name, make in (("direct", lambda r: DirectPopulation.random(203, r)), ("network", lambda r: NetworkPopulation.random(101, r))):
        rng = np.random.default_rng(12)
        ev = sharded_evaluator(lambda block: block.body_counts(1).astype(np.float64) * 1.5, n_threads=1)
        pop, fit, hist = run_generations(make(rng), 2, ev, rng, 0.2, 0.2, 0.2)
        np.save(os.path.join(out_dir, "%s_fit%d.npy" % (name, rank)), fit)
        np.save(os.path.join(out_dir, "%s_cost%d.npy" % (name, rank)), ev.last_shard_cost)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_generations_of_the_direct_and_network_populations_gloo(tmp_path, built):
    """World size 2 over gloo: replicated variation, cost-balanced shards, one fitness all-gather per generation -- every rank
    ends with the single-process fitness, and the ranks' predicted costs are within 5 % of each other."""
    import os
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for name, make in (("direct", lambda r: DirectPopulation.random(203, r)), ("network", lambda r: NetworkPopulation.random(101, r))):
        rng = np.random.default_rng(12)
        pop, fit, hist = run_generations(make(rng), 2, lambda p: p.body_counts(1).astype(np.float64) * 1.5, rng, 0.2, 0.2, 0.2)
        for r in range(2):
            assert np.array_equal(np.load(os.path.join(str(tmp_path), "%s_fit%d.npy" % (name, r))), fit), name
            cost = np.load(os.path.join(str(tmp_path), "%s_cost%d.npy" % (name, r)))
            assert cost.sum() == pop.body_counts(1).sum() and abs(cost[0] - cost[1]) <= 0.05 * cost.mean(), (name, cost)
