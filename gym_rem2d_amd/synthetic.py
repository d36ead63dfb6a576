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
