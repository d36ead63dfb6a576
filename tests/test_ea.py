"""EA loop restatement (REM2D_main.py:241-348) -- host logic on CPU, one real generation on the GPU."""
import copy
import random

import numpy as np
import pytest

from gym_rem2d_amd import ea


def test_tournament_prefers_fitter():
    random.seed(0)
    pop = []
    for f in range(20):
        ind = ea.Individual()
        ind.fitness = float(f)
        pop.append(ind)
    chosen = ea.sel_tournament(pop, 2000)
    assert len(chosen) == 2000
    assert np.mean([c.fitness for c in chosen]) > 14.0  # E[max of 4 uniform draws from 0..19] = 15.3
    assert max(c.fitness for c in chosen) == 19.0


def test_generation_loop_with_stub_evaluator(tmp_path):
    """Fitness = number of nodes: selection + mutation must grow creatures over generations."""
    cfg = ea.make_config(population_size=24, morphmutation_prob=0.3, mutation_prob=0.1, encoding="direct",
                         checkpoint_frequency=2)
    calls = []

    def evaluate_batch(inds):
        calls.append(len(inds))
        return [float(len(i.genome.create(7).getNodes())) for i in inds]

    pop, hist = ea.run_ea(cfg, evaluate_batch=evaluate_batch, seed=3, n_generations=6, save_dir=str(tmp_path), log=None)
    assert calls == [24] * 7 and len(pop) == 24 and len(hist) == 6
    assert hist[-1][3] >= hist[0][3]               # mean fitness does not collapse
    assert (tmp_path / "s_pop0").exists() and (tmp_path / "s_pop5").exists() and (tmp_path / "s_elite5").exists()
    # the files are in the reference's pickle format (class paths of REM2D_main / Encodings / gym_rem2D ...)
    from gym_rem2d_amd.compat import FitnessData, load_reference_pickle
    best = load_reference_pickle(str(tmp_path / "s_elite5"))
    assert best.fitness == max(p.fitness for p in pop)
    saved = load_reference_pickle(str(tmp_path / "s_pop5"))
    assert [i.fitness for i in saved] == [i.fitness for i in pop]
    fd = load_reference_pickle(str(tmp_path / "s_"))
    assert isinstance(fd, FitnessData) and len(fd.avg) == 6 and fd.p_100[-1] == hist[-1][2] and fd.p_0[-1] == hist[-1][1]
    import pickletools
    names = {a for op, a, _ in pickletools.genops(open(tmp_path / "s_pop5", "rb").read()) if op.name == "GLOBAL"}
    assert "REM2D_main Individual" in names and "Encodings.direct_encoding DirectEncoding" in names
    assert not any(n.startswith("gym_rem2d_amd") for n in names)


def test_individual_random_matches_module_defaults():
    random.seed(5)
    ind = ea.Individual.random(encoding="lsystem")
    assert ind.tree_depth == 8 and len(ind.genome.moduleList) == 8
    clone = copy.deepcopy(ind)
    clone.mutate(0.5, 0.5, 0.5)
    assert ind.genome is not clone.genome
    with pytest.raises(Exception, match="cellular"):
        ea.Individual.random(encoding="ce")
    net = ea.Individual.random(encoding="cppn")
    nodes = net.genome.create(7).getNodes()
    assert nodes[0].parent == -1 and nodes[0].type == -1 and len(nodes) <= 22  # axiom moduleRef -1 as in the reference
    for n in nodes[1:]:
        assert n.parent_connection_coordinates is not None and n.controller is not None
        assert 0.5 <= getattr(n.module_, "width", 0.5) <= 1.0


@pytest.mark.gpu
def test_one_generation_on_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = ea.make_config(population_size=32, encoding="lsystem")
    pop, hist = ea.run_ea(cfg, seed=1, n_generations=2, log=None)
    fits = [p.fitness for p in pop]
    assert len(fits) == 32 and all(np.isfinite(fits)) and max(fits) > 0
    assert hist[-1][2] == max(fits)


def test_parallel_encoder_equals_serial_path():
    """encode_population on a fork pool == the serial compiler path, creature for creature and in the same
    (lane bucket, pipeline period, rounds, bodies) order as BatchedModular2D.reset_specs."""
    import random
    import numpy as np
    from gym_rem2d_amd import Morphology, build_creature
    from gym_rem2d_amd.compiler import lanes_for
    from gym_rem2d_amd.ea import Individual
    import gym_rem2d_amd.encode as E
    random.seed(5)
    inds = [Individual.random(encoding="lsystem" if k % 3 else "direct") for k in range(2300)]
    serial = E.encode_population(inds, 8, workers=1)
    pooled = E.encode_population(inds, 8, workers=3, chunk=200)
    import copy
    specs = [build_creature(copy.deepcopy(ind.genome.create(8)).getNodes(), ind.genome.moduleList)[0] for ind in inds]
    groups = {}
    for e, s in enumerate(specs):
        groups.setdefault(lanes_for(s.n_bodies), []).append(e)
    assert len(serial) == len(pooled) == len(groups)
    for (m1, i1), (m2, i2), lanes in zip(serial, pooled, sorted(groups)):
        ref = sorted(groups[lanes], key=lambda e: (specs[e].period, max(specs[e].rounds, default=-1), specs[e].n_bodies))
        assert i1 == i2 == ref
        mr = Morphology.from_specs([specs[e] for e in ref], lanes)
        for k in mr.arrays:
            assert np.array_equal(m1.arrays[k], mr.arrays[k]) and np.array_equal(m2.arrays[k], mr.arrays[k])
        assert np.array_equal(m1.n_bodies, mr.n_bodies) and np.array_equal(m2.n_bodies, mr.n_bodies)


@pytest.mark.gpu
@pytest.mark.parametrize("encoding", ["lsystem", "direct", "cppn"])
def test_generation_on_gpu_equals_generation_with_the_oracle(encoding, oracle):
    """(f2) The EA generation of REM2D_main.py:280-298 -- tournament-4, clone, mutate, evaluate -- run twice from the
    same seed: once with the batched GPU episode as the evaluator, once with the CPU oracle.  Fitness feeds selection,
    so any difference in any fitness bit would change the offspring of the next generation: populations, fitness
    lists and per-generation statistics must be identical."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gym_rem2d_amd import _lib, make_terrain
    from gym_rem2d_amd.encode import encode_population
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import evaluate_population
    steps = 500
    cfg = ea.make_config(population_size=40, encoding=encoding, morphmutation_prob=0.2, mutation_prob=0.2)
    depth = int(cfg["morphology"]["max_depth"])
    env = BatchedModular2D(flags=_lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN)

    def gpu_eval(inds):
        return evaluate_population(inds, tree_depth=depth, env=env, max_steps=steps)

    terrain = make_terrain(4)
    xs, ys, _ = terrain.f32()
    ot = oracle.Terrain(xs, ys, None, terrain.friction)

    def oracle_eval(inds):
        out = [0.0] * len(inds)
        for m, idx in encode_population(inds, depth, workers=1):
            fit = oracle.batch_run(ot, m.as_dict(), steps, n_threads=8, flags=oracle.FLAG_CONTINUOUS)["fitness"]
            for e, f in zip(idx, fit):
                out[e] = float(f)
        return out

    pop_g, hist_g = ea.run_ea(cfg, evaluate_batch=gpu_eval, seed=21, n_generations=3, log=None)
    pop_o, hist_o = ea.run_ea(cfg, evaluate_batch=oracle_eval, seed=21, n_generations=3, log=None)
    env.close()
    assert [p.fitness for p in pop_g] == [p.fitness for p in pop_o]
    assert [h[:4] for h in hist_g] == [h[:4] for h in hist_o]
    assert len({p.fitness for p in pop_g}) > 5 and max(p.fitness for p in pop_g) > 0
    for a, b in zip(pop_g, pop_o):   # same genomes came out of selection + mutation
        na, nb = a.genome.create(depth).getNodes(), b.genome.create(depth).getNodes()
        assert [(n.index, n.parent, n.type) for n in na] == [(n.index, n.parent, n.type) for n in nb]
