"""Native genotype -> phenotype compiler (rem2d_compile_lsystem, host C++) vs the Python compiler, which is
itself pinned bit-for-bit against the reference's fixtures (tests/test_host_golden.py): every output word of
every creature must be identical."""
import random

import numpy as np
import pytest

from gym_rem2d_amd import get_module_list
from gym_rem2d_amd.encodings import LSystem


class _Ind:
    tree_depth = 8

    def __init__(self, genome):
        self.genome = genome


def _population(n, max_modules, mutate=True, seed0=0):
    inds = []
    for seed in range(seed0, seed0 + n):
        random.seed(seed)
        g = LSystem(get_module_list())
        g.maxModules = max_modules
        if mutate and seed % 2 == 1:
            for _ in range(1 + seed % 4):
                g.mutate(0.5, 0.5, 0.5)
        inds.append(_Ind(g))
    return inds


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from gym_rem2d_amd import encode
    return encode


@pytest.mark.parametrize("max_modules,n", [(15, 1500), (20, 600), (40, 400), (3, 100)])
def test_native_compiler_equals_python_compiler(built, max_modules, n):
    inds = _population(n, max_modules)
    py = built.encode_population(inds, 8, workers=1)
    nat = built.encode_lsystem_native(inds, n_threads=3)
    assert [b[0].lanes for b in py] == [b[0].lanes for b in nat]
    for (mp, ip), (mn, in_) in zip(py, nat):
        assert ip == in_
        assert np.array_equal(mp.n_bodies, mn.n_bodies)
        for k in mp.arrays:
            assert np.array_equal(mp.arrays[k], mn.arrays[k]), (k, max_modules)
    # bodies actually vary: the comparison is not vacuous
    assert sum(len(b[1]) for b in nat) == n and len(nat) >= 2


def test_native_compiler_single_thread_and_errors(built):
    inds = _population(64, 15, seed0=5000)
    a = built.encode_lsystem_native(inds, n_threads=1)
    b = built.encode_lsystem_native(inds, n_threads=8)
    for (m1, i1), (m2, i2) in zip(a, b):
        assert i1 == i2 and all(np.array_equal(m1.arrays[k], m2.arrays[k]) for k in m1.arrays)
    from gym_rem2d_amd import _lib
    arrays = built.lsystem_genome_arrays([i.genome for i in _population(40, 40)])
    with pytest.raises(_lib.Rem2dError, match="more bodies than"):
        built.compile_lsystem_arrays(arrays, 8, 40, 4)     # 41-node trees do not fit 4 lanes


def test_array_population_round_trip_and_operators(built):
    """LSystemPopulation: arrays <-> object genomes express to the same creatures; mutation keeps every
    invariant of the object genomes (ranges, distinct sites, rule sizes); selection clones."""
    from gym_rem2d_amd.population import LSystemPopulation, tournament
    rng = np.random.default_rng(3)
    pop = LSystemPopulation.random(700, rng, max_modules=20)
    for _ in range(6):
        pop.mutate(0.3, 0.3, 0.2, rng)
    a = pop.a
    n_box = int((a["mod_shape"][0] == 1).sum())
    assert a["rule_n"].min() >= 0 and a["rule_n"][:, :n_box].max() <= 3 and a["rule_n"][:, n_box:].max() == 0
    assert np.all(np.sort(a["rule_site"], axis=2) == np.array([0, 1, 2]))          # sites stay a permutation
    assert np.all((a["mod_width"][:, :n_box] >= 0.5) & (a["mod_width"][:, :n_box] <= 1.0))
    assert np.all((a["mod_radius"][:, n_box:] >= 0.25) & (a["mod_radius"][:, n_box:] <= 0.5))
    assert np.all(np.abs(a["ctl_offset"]) <= a["mod_angle"] / 2 + 1e-15) and np.all((a["ctl_amp"] >= 0) & (a["ctl_amp"] <= 1))
    # arrays -> objects -> python compiler == arrays -> native compiler
    genomes = pop.to_genomes()
    py = built.encode_population([_Ind(g) for g in genomes], 8, workers=1)
    nat = pop.compile(n_threads=2)
    assert len(py) == len(nat)
    for (mp, ip), (mn, in_) in zip(py, nat):
        assert ip == in_ and all(np.array_equal(mp.arrays[k], mn.arrays[k]) for k in mp.arrays)
    # objects -> arrays is the inverse
    back = LSystemPopulation.from_genomes(genomes)
    for k in ("mod_shape", "mod_width", "mod_height", "mod_radius", "mod_angle", "ctl_amp", "ctl_offset", "rule_n"):
        assert np.array_equal(back.a[k], a[k]), k
    live = np.arange(3)[None, None, :] < a["rule_n"][:, :, None]
    assert np.array_equal(np.where(live, back.a["rule_site"], 0), np.where(live, a["rule_site"], 0))
    assert np.array_equal(np.where(live, back.a["rule_ref"], 0), np.where(live, a["rule_ref"], 0))
    # selection: winners have the best fitness of their 4 aspirants on average -> mean goes up; clones are copies
    fit = rng.random(700)
    idx = tournament(fit, 700, rng)
    assert fit[idx].mean() > fit.mean() + 0.2
    sel = pop.select(idx)
    assert np.array_equal(sel.a["mod_angle"], a["mod_angle"][idx])
    before = a["mod_angle"].copy()
    sel.a["mod_angle"][:] = -1
    assert np.array_equal(a["mod_angle"], before)


@pytest.mark.parametrize("encoding,n", [("direct", 700), ("cppn", 250), ("lsystem", 300), ("mixed", 300)])
def test_native_tree_compiler_equals_python_compiler(built, encoding, n):
    """rem2d_compile_trees (any encoding: the phenotype tree node by node) vs the Python compiler, word for word:
    direct genomes (Direct_Encoding.py:18-27), network genomes (Network_Encoding.py:86-139,171-206), L-system trees
    taken the long way round, and a mixed population."""
    from gym_rem2d_amd.ea import Individual
    random.seed(77)
    kinds = {"mixed": ["direct", "lsystem", "cppn"]}.get(encoding, [encoding])
    inds = [Individual.random(encoding=kinds[k % len(kinds)]) for k in range(n)]
    for k, ind in enumerate(inds):
        if k % 3 == 0:
            for _ in range(1 + k % 3):
                ind.mutate(0.4, 0.4, 0.3)
    depth = 7
    py = built.encode_population(inds, depth, workers=1)
    nat = built.encode_trees_native(inds, depth, n_threads=3)
    assert [b[0].lanes for b in py] == [b[0].lanes for b in nat]
    for (mp, ip), (mn, in_) in zip(py, nat):
        assert ip == in_
        assert np.array_equal(mp.n_bodies, mn.n_bodies)
        for k in mp.arrays:
            assert np.array_equal(mp.arrays[k], mn.arrays[k]), (k, encoding)
    assert sum(len(b[1]) for b in nat) == n
    assert max(int(b[0].n_bodies.max()) for b in nat) >= 3


def test_native_tree_compiler_errors(built):
    from gym_rem2d_amd import _lib
    from gym_rem2d_amd.ea import Individual
    random.seed(3)
    inds = [Individual.random(encoding="direct") for _ in range(20)]
    trees = [i.genome.create(7) for i in inds]
    arrays = built.tree_batch_arrays(trees, [i.genome.moduleList for i in inds])
    assert arrays["node_count"].max() >= 3
    with pytest.raises(_lib.Rem2dError, match="more bodies"):
        built.compile_tree_arrays(arrays, 2)           # 2 lanes cannot hold the larger creatures
    with pytest.raises(ValueError, match="more than max_nodes"):
        built.tree_batch_arrays(trees, None, max_nodes=2)
    bad = dict(arrays)
    bad["node_count"] = arrays["node_count"].copy()
    bad["node_count"][0] = 1000
    with pytest.raises(_lib.Rem2dError, match="node_count"):
        built.compile_tree_arrays(bad, 32)


@pytest.mark.parametrize("depth", [7, 4])
def test_native_network_expansion_equals_python(built, depth):
    """rem2d_compile_network: the NN-query tree growth of the network encoding (Network_Encoding.py:86-139,171-206:
    update / iterate / create / recursiveNodeGen, setMorph, setControl) + create_robot, all native, vs the Python
    path word for word -- incl. mutated prototypes and mutated networks."""
    from gym_rem2d_amd.ea import Individual
    random.seed(123)
    inds = [Individual.random(encoding="cppn") for _ in range(400)]
    for k, ind in enumerate(inds):
        for _ in range(k % 4):
            ind.mutate(0.4, 0.4, 0.3)
    py = built.encode_population(inds, depth, workers=1)
    nat = built.encode_network_native(inds, depth, n_threads=3)
    assert [b[0].lanes for b in py] == [b[0].lanes for b in nat]
    for (mp, ip), (mn, in_) in zip(py, nat):
        assert ip == in_
        assert np.array_equal(mp.n_bodies, mn.n_bodies)
        for k in mp.arrays:
            assert np.array_equal(mp.arrays[k], mn.arrays[k]), (k, depth)
    sizes = np.concatenate([b[0].n_bodies for b in nat])
    assert len(sizes) == 400 and sizes.max() >= 6 and len(np.unique(sizes)) >= 4   # the population is varied
