"""Synthetic populations for the BASELINE.json configurations.

All inputs are synthetic and generated on the host with seeded Python/numpy RNGs:
  config 1: one random direct-encoding individual (random.seed(0))
  config 2: identical 4-module chain creatures (boxes 0.5 x 0.8, top sites, sine controllers)
  config 3: random L-system creatures, padded to 16 lanes
"""
import copy
import math
import random

import numpy as np

from .compiler import Morphology, build_creature, lanes_for
from .controller import Controller
from .encodings import DirectEncoding, LSystem, NNEncoding
from .modules import BoxConnection, Standard2D, get_module_list
from .tree import Node, Tree


def chain_tree(n_modules=4, site="top", width=0.5, height=0.8, angle=math.pi / 2, amp=0.5, freq=0.1):
    """boxes chained on `site`, controllers amp 0.5, freq 0.1 rad/step,
    phase -1 + 2k/3, offset 0."""
    state = random.getstate()
    tree = Tree([])
    for k in range(n_modules):
        mod = Standard2D()
        mod.width, mod.height, mod.angle = width, height, angle
        c = Controller()
        c.amplitude, c.frequency, c.phase, c.offset = amp, freq, -1 + 2 * k / 3, 0.0
        con = None if k == 0 else getattr(BoxConnection, site)
        tree.nodes.append(Node(k, k - 1, 0, con, c, module_=mod))
    random.setstate(state)  # building the fixed chain must not disturb seeded streams
    return tree


def spec_from_tree(tree, module_list=None):
    t = copy.deepcopy(tree)
    spec, _, _ = build_creature(t.getNodes(), module_list if module_list is not None else [0])
    return spec


def chain_population(n_envs, n_modules=4, site="top", lanes=None):
    spec = spec_from_tree(chain_tree(n_modules, site))
    return Morphology.replicate(spec, n_envs, lanes)


def lsystem_specs(seeds, max_modules=15, mutate_odd=False):
    """One L-system creature per seed (config 3: maxModules=15 so that <= 16 nodes)."""
    specs = []
    for seed in seeds:
        random.seed(int(seed))
        ml = get_module_list()
        g = LSystem(ml)
        g.maxModules = max_modules
        if mutate_odd and seed % 2 == 1:
            for _ in range(3):
                g.mutate(0.5, 0.5, 0.5)
        tree = g.create(8)
        spec, _, _ = build_creature(tree.getNodes(), ml)
        specs.append(spec)
    return specs


def direct_specs(seeds):
    specs = []
    for seed in seeds:
        random.seed(int(seed))
        ml = get_module_list()
        g = DirectEncoding(ml)
        tree = copy.deepcopy(g.create(8))
        tree.getNodes()
        spec, _, _ = build_creature(tree.getNodes(), ml)
        specs.append(spec)
    return specs


def lsystem_population(n_envs, lanes=16, n_unique=None, seed0=0):
    """n_envs L-system creatures; with n_unique < n_envs the unique set is tiled (host build
    time is O(n_unique))."""
    n_unique = n_envs if n_unique is None else min(n_unique, n_envs)
    specs = lsystem_specs(range(seed0, seed0 + n_unique))
    uniq = Morphology.from_specs(specs, lanes)
    if n_unique == n_envs:
        return uniq
    idx = np.arange(n_envs) % n_unique
    return uniq.take(idx)


def cppn_specs(seeds):
    """Config 4 input: one network-encoded creature per seed (synthetic feed-forward CPPN genome)."""
    specs = []
    for seed in seeds:
        random.seed(int(seed))
        ml = get_module_list()
        g = NNEncoding(ml)
        tree = g.create(7)
        spec, _, _ = build_creature(tree.getNodes(), g.moduleList)
        specs.append(spec)
    return specs


# ------------------------------------------------------------------------------------------------
# the same synthetic populations through the native compilers (rem2d_compile_lsystem / rem2d_compile_network): only the
# genome -- a few dozen numbers drawn from `random` after random.seed(seed), the part that defines "seed k's creature" --
# is made in Python; tree growth, create_robot, schedule and SoA packing are native (3 us instead of 0.4 ms a creature).
# tests/test_bench_host.py: the batches equal Morphology.from_specs(lsystem_specs / cppn_specs) word for word.
# ------------------------------------------------------------------------------------------------
def _lsystem_genome_chunk(args):
    seeds, max_modules = args
    from . import encode
    gs = []
    for seed in seeds:
        random.seed(int(seed))
        g = LSystem(get_module_list())
        g.maxModules = max_modules
        gs.append(g)
    return encode.lsystem_genome_arrays(gs)


def _cppn_genome_chunk(args):
    seeds, _ = args
    from . import encode
    gs = []
    for seed in seeds:
        random.seed(int(seed))
        gs.append(NNEncoding(get_module_list()))
    return encode.network_genome_arrays(gs)


def _genome_arrays(maker, seeds, extra, n_proc):
    """Genome arrays of `seeds`, made by up to n_proc forked workers (the caller must not have initialised the GPU)."""
    seeds = np.asarray(list(seeds), dtype=np.int64)
    if len(seeds) == 0:
        raise ValueError("no seeds")
    n_proc = max(1, min(int(n_proc), len(seeds) // 256 or 1))
    chunks = [c.tolist() for c in np.array_split(seeds, n_proc * 4 if n_proc > 1 else 1) if len(c)]
    if n_proc > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(n_proc) as pool:
            parts = pool.map(maker, [(c, extra) for c in chunks])
    else:
        parts = [maker((c, extra)) for c in chunks]
    return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}


def bucket_batches(m, descending=True):
    """A compiled batch on wide lanes -> one Morphology per lane count (2, 4, 8, ...), each sorted by (pipeline period,
    joint rounds, bodies, original index) -- the most complex creatures first by default: their wavefronts are the long
    ones and should be dispatched first.  Returns [(Morphology, indices into m)]."""
    K = m.lanes
    nb = m.n_bodies.astype(np.int64)
    lanes_of = np.array([lanes_for(int(v)) for v in range(int(nb.max()) + 1)], dtype=np.int64)[nb]
    jr = m.arrays["jround"].reshape(m.n_envs, K)
    period = np.maximum(1, (jr >> 16).max(axis=1) & 0xff).astype(np.int64)
    has_joint = m.arrays["parent"].reshape(m.n_envs, K) >= 0
    max_round = np.where(has_joint, jr & 0xff, -1).max(axis=1).astype(np.int64)
    out = []
    for lanes in sorted(set(lanes_of.tolist())):
        idx = np.nonzero(lanes_of == lanes)[0]
        # a stable sort on the key triple, like list.sort(key=(period, max round, bodies), reverse=descending)
        key = (period[idx] << 32) | ((max_round[idx] + 1) << 16) | nb[idx]
        order = np.argsort(-key if descending else key, kind="stable")
        idx = idx[order]
        b = Morphology(len(idx), lanes)
        for k, v in m.arrays.items():
            b.arrays[k][:] = v.reshape(m.n_envs, K)[idx, :lanes].reshape(-1)
        b.n_bodies[:] = m.n_bodies[idx]
        out.append((b, idx))
    return out


def lsystem_batches_native(seeds, max_modules=15, n_proc=1, n_threads=0, descending=True):
    """Config 3's population (one L-system creature per seed, ``lsystem_specs``) as lane buckets, natively compiled."""
    from . import encode
    arrays = _genome_arrays(_lsystem_genome_chunk, seeds, max_modules, n_proc)
    lanes = 64 if max_modules + 1 > 32 else lanes_for(max_modules + 1)
    m = encode.compile_lsystem_arrays(arrays, 8, max_modules, lanes, n_threads)   # LSystem.treeDepth = 8 (LSystem.py:134-137)
    return bucket_batches(m, descending)


def cppn_batches_native(seeds, n_proc=1, n_threads=0, descending=True):
    """Config 4's population (one network-encoded creature per seed, ``cppn_specs``) as lane buckets, natively compiled."""
    from . import encode
    arrays = _genome_arrays(_cppn_genome_chunk, seeds, None, n_proc)
    m = encode.compile_network_arrays(arrays, 7, 20, 32, n_threads)   # NNEncoding defaults: depth 7, maxModules 20
    return bucket_batches(m, descending)
