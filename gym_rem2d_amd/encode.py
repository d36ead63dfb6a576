"""Genotype -> phenotype -> SoA morphology for a whole population, on all host cores.

SURVEY.md 8f rank 1: once the episode runs on the GPU, ``genome.create()`` + ``create_robot`` (Python, ~0.5 ms per
individual) is what a generation waits for -- 65 s for the 131 072 individuals one GPU evaluates in 4 s.  The
work is independent per individual, so it is cut into chunks for a fork pool: the workers inherit the
population (nothing is pickled on the way in), build trees and creatures with the same code as the serial
path (`compiler.build_creature`, bit-exact vs the reference fixtures), pack them per lane count into
`Morphology` arrays and send back only those arrays plus the per-creature sort keys.  The workers never touch
the GPU.
"""
import copy
import multiprocessing as mp
import os

import numpy as np

from .compiler import Morphology, build_creature, lanes_for

_POP = None  # population visible to forked workers


def _encode_range(args):
    lo, hi, tree_depth = args
    specs = []
    for ind in _POP[lo:hi]:
        # like Modular2D.reset (Modular2DEnv.py:568) work on a copy: create_robot marks nodes as expressed,
        # and a direct encoding hands out the tree it keeps
        tree = copy.deepcopy(ind.genome.create(tree_depth if tree_depth is not None else ind.tree_depth))
        specs.append(build_creature(tree.getNodes(), ind.genome.moduleList)[0])
    groups = {}
    for k, s in enumerate(specs):
        groups.setdefault(lanes_for(s.n_bodies), []).append(k)
    out = []
    for lanes, ks in groups.items():
        m = Morphology.from_specs([specs[k] for k in ks], lanes)
        keys = np.array([(specs[k].period, max(specs[k].rounds, default=-1), specs[k].n_bodies) for k in ks], dtype=np.int64)
        out.append((lanes, m.arrays, m.n_bodies, np.asarray(ks, dtype=np.int64) + lo, keys.reshape(-1, 3)))
    return out


def encode_population(individuals, tree_depth=None, workers=None, chunk=256):
    """Returns the ``batches`` list ``BatchedModular2D._upload`` takes: [(Morphology, env indices)], one per
    lane count, creatures of a bucket sorted by (pipeline period, joint rounds, bodies) so that the waves of
    the step kernel are homogeneous."""
    global _POP
    n = len(individuals)
    if workers is None:
        workers = min(os.cpu_count() or 1, 64)
        try:   # forking a process that holds a GPU context costs ~0.2 s per worker
            import torch
            if torch.cuda.is_initialized():
                workers = min(workers, 8)
        except Exception:  # noqa: BLE001
            pass
    ranges = [(lo, min(n, lo + chunk), tree_depth) for lo in range(0, n, chunk)]
    _POP = individuals
    try:
        if workers <= 1 or n < 2048:   # a fork pool costs ~0.3 s to start: not worth it for small populations
            parts = [_encode_range(r) for r in ranges]
        else:
            with mp.get_context("fork").Pool(min(workers, len(ranges))) as pool:
                parts = pool.map(_encode_range, ranges)
    finally:
        _POP = None
    by_lanes = {}
    for part in parts:
        for lanes, arrays, n_bodies, idx, keys in part:
            by_lanes.setdefault(lanes, []).append((arrays, n_bodies, idx, keys))
    batches = []
    for lanes in sorted(by_lanes):
        items = by_lanes[lanes]
        idx = np.concatenate([it[2] for it in items])
        keys = np.concatenate([it[3] for it in items])
        order = np.lexsort((idx, keys[:, 2], keys[:, 1], keys[:, 0]))   # stable: ties keep population order
        m = Morphology(len(idx), lanes)
        lane_order = (order[:, None] * lanes + np.arange(lanes)[None, :]).reshape(-1)
        for k in m.arrays:
            m.arrays[k][:] = np.concatenate([it[0][k] for it in items])[lane_order]
        m.n_bodies[:] = np.concatenate([it[1] for it in items])[order]
        batches.append((m, idx[order].tolist()))
    return batches


# ------------------------------------------------------------------------------------------------
# native path for L-system genomes: rem2d_compile_lsystem (csrc/rem2d_compile.h), all host threads
# ------------------------------------------------------------------------------------------------
_SITE = {"left": 0, "right": 1, "top": 2}


def lsystem_genome_arrays(genomes):
    """SoA view of a list of ``LSystem`` genomes (the input of rem2d_compile_lsystem)."""
    n = len(genomes)
    T = len(genomes[0].moduleList) if n else 0
    a = dict(mod_shape=np.zeros((n, T), np.int32), rule_n=np.zeros((n, T), np.int32),
             rule_site=np.zeros((n, T, 3), np.int32), rule_ref=np.zeros((n, T, 3), np.int32))
    for k in ("mod_width", "mod_height", "mod_radius", "mod_angle", "mod_torque", "ctl_amp", "ctl_phase", "ctl_freq",
              "ctl_offset"):
        a[k] = np.zeros((n, T), np.float64)
    for e, g in enumerate(genomes):
        if len(g.moduleList) != T or len(g.rules) != T:
            raise ValueError("all genomes must have the same number of module types")
        for t, m in enumerate(g.moduleList):
            box = m.type == "SIMPLE"
            a["mod_shape"][e, t] = 1 if box else 2
            if box:
                a["mod_width"][e, t], a["mod_height"][e, t] = m.width, m.height
            else:
                a["mod_radius"][e, t] = m.radius
            a["mod_angle"][e, t], a["mod_torque"][e, t] = m.angle, m.torque
            c = m.controller
            a["ctl_amp"][e, t], a["ctl_phase"][e, t], a["ctl_freq"][e, t], a["ctl_offset"][e, t] = \
                c.amplitude, c.phase, c.frequency, c.offset
            kids = g.rules[t].module.children
            a["rule_n"][e, t] = len(kids)
            for k, s in enumerate(kids):
                a["rule_site"][e, t, k] = _SITE[s.parentConnectionSite.name]
                a["rule_ref"][e, t, k] = s.moduleRef
    return a


def compile_lsystem_arrays(arrays, tree_depth, max_modules, lanes, n_threads=0):
    """rem2d_compile_lsystem on genome arrays -> (Morphology with `lanes` lanes per creature, n_bodies)."""
    import ctypes as C
    from . import _lib
    from .compiler import TERRAIN_HEIGHT
    n = int(arrays["mod_shape"].shape[0])
    T = int(arrays["mod_shape"].shape[1])
    m = Morphology(n, lanes)
    G = _lib.LsystemGenomes()
    G.n, G.n_types = n, T
    keep = []
    for k in ("mod_shape", "mod_width", "mod_height", "mod_radius", "mod_angle", "mod_torque", "ctl_amp", "ctl_phase",
              "ctl_freq", "ctl_offset", "rule_n", "rule_site", "rule_ref"):
        v = np.ascontiguousarray(arrays[k])
        keep.append(v)
        setattr(G, k, v.ctypes.data_as(C.c_void_p))
    out = _lib.Morph()
    for k in m.arrays:
        setattr(out, k, m.arrays[k].ctypes.data_as(C.c_void_p))
    _lib.check(_lib.lib().rem2d_compile_lsystem(C.byref(G), int(tree_depth), int(max_modules), float(TERRAIN_HEIGHT),
                                                int(lanes), C.byref(out), m.n_bodies.ctypes.data_as(C.c_void_p),
                                                int(n_threads)))
    return m


def count_lsystem_bodies(arrays, tree_depth, max_modules, n_threads=0):
    """rem2d_compile_lsystem with out = NULL: the body count of every genome's creature (expression + create_robot's skip rules,
    nothing written) -- int32 [n].  A static cost key: lanes, joint rounds and solver work all grow with it."""
    import ctypes as C
    from . import _lib
    from .compiler import TERRAIN_HEIGHT
    n, T = int(arrays["mod_shape"].shape[0]), int(arrays["mod_shape"].shape[1])
    G = _lib.LsystemGenomes()
    G.n, G.n_types = n, T
    keep = []
    for k in ("mod_shape", "mod_width", "mod_height", "mod_radius", "mod_angle", "mod_torque", "ctl_amp", "ctl_phase",
              "ctl_freq", "ctl_offset", "rule_n", "rule_site", "rule_ref"):
        v = np.ascontiguousarray(arrays[k])
        keep.append(v)
        setattr(G, k, v.ctypes.data_as(C.c_void_p))
    nb = np.zeros(n, dtype=np.int32)
    _lib.check(_lib.lib().rem2d_compile_lsystem(C.byref(G), int(tree_depth), int(max_modules), float(TERRAIN_HEIGHT), 64, None,
                                                nb.ctypes.data_as(C.c_void_p), int(n_threads)))
    return nb


def batches_from_compiled(m):
    """Split a wide compiled batch into per-lane-count buckets, sorted like encode_population."""
    K = m.lanes
    nb = m.n_bodies.astype(np.int64)
    want = np.array([lanes_for(int(v)) for v in np.unique(nb)])
    lut = dict(zip(np.unique(nb).tolist(), want.tolist()))
    lanes_of = np.array([lut[int(v)] for v in nb], dtype=np.int64)
    jr = m.arrays["jround"].reshape(m.n_envs, K)
    period = (jr[:, 0] >> 16) & 0xff
    has_joint = m.arrays["parent"].reshape(m.n_envs, K) >= 0
    max_round = np.where(has_joint, jr & 0xff, -1).max(axis=1)
    batches = []
    for lanes in sorted(set(lanes_of.tolist())):
        idx = np.nonzero(lanes_of == lanes)[0]
        order = np.lexsort((idx, nb[idx], max_round[idx], period[idx]))
        idx = idx[order]
        b = Morphology(len(idx), lanes)
        for k, v in m.arrays.items():
            b.arrays[k][:] = v.reshape(m.n_envs, K)[idx, :lanes].reshape(-1)
        b.n_bodies[:] = m.n_bodies[idx]
        batches.append((b, idx.tolist()))
    return batches


def encode_lsystem_native(individuals, tree_depth=None, n_threads=0):
    """Same result as encode_population for L-system genomes, through the native compiler."""
    genomes = [ind.genome for ind in individuals]
    if not genomes:
        return []
    arrays = lsystem_genome_arrays(genomes)
    depth = genomes[0].treeDepth          # LSystem.create ignores its argument (LSystem.py:144-151)
    maxm = genomes[0].maxModules
    if any(g.treeDepth != depth or g.maxModules != maxm for g in genomes):
        raise ValueError("genomes of one batch must share treeDepth / maxModules")
    lanes = 64 if maxm + 1 > 32 else lanes_for(maxm + 1)
    return batches_from_compiled(compile_lsystem_arrays(arrays, depth, maxm, lanes, n_threads))


# ------------------------------------------------------------------------------------------------
# native path for every other encoding: flatten the phenotype trees, rem2d_compile_trees does the rest
# ------------------------------------------------------------------------------------------------
_TREE_F64 = ("width", "height", "radius", "angle", "torque", "ctl_amp", "ctl_phase", "ctl_freq", "ctl_offset")
_TREE_I32 = ("index", "parent", "site", "shape")


def tree_batch_arrays(trees, module_lists=None, max_nodes=None):
    """Flatten phenotype trees (``genome.create()`` of any encoding) into the [n][max_nodes] node tables of
    rem2d_compile_trees: per node its index / parent / connection site and the parameters of ITS module and controller
    (``node.module_`` where the encoding set one -- Direct_Encoding.py:18-27, Network_Encoding.py:97-129 -- else the
    prototype ``module_list[node.type]``).  Attribute reads only: no deepcopy, no trigonometry."""
    n = len(trees)
    node_lists = [t.getNodes() for t in trees]
    if max_nodes is None:
        max_nodes = max([len(nl) for nl in node_lists] + [1])
    a = {k: np.zeros((n, max_nodes), np.int32) for k in _TREE_I32}
    a.update({k: np.zeros((n, max_nodes), np.float64) for k in _TREE_F64})
    a["site"][:] = -1
    count = np.zeros(n, np.int32)
    idx, par, site, shape = a["index"], a["parent"], a["site"], a["shape"]
    wid, hei, rad, ang, tor = a["width"], a["height"], a["radius"], a["angle"], a["torque"]
    amp, pha, fre, off = a["ctl_amp"], a["ctl_phase"], a["ctl_freq"], a["ctl_offset"]
    for e, nodes in enumerate(node_lists):
        if len(nodes) > max_nodes:
            raise ValueError("tree %d has %d nodes, more than max_nodes=%d" % (e, len(nodes), max_nodes))
        ml = module_lists[e] if module_lists is not None else None
        count[e] = len(nodes)
        for i, nd in enumerate(nodes):
            m = nd.module_ if nd.module_ is not None else ml[nd.type]
            idx[e, i], par[e, i] = nd.index, nd.parent
            con = nd.parent_connection_coordinates
            if con is not None:
                site[e, i] = _SITE[con.name]
            if m.type == "SIMPLE":
                shape[e, i], wid[e, i], hei[e, i] = 1, m.width, m.height
            else:
                shape[e, i], rad[e, i] = 2, m.radius
            ang[e, i], tor[e, i] = m.angle, m.torque
            c = nd.controller
            if c is not None:
                amp[e, i], pha[e, i], fre[e, i], off[e, i] = c.amplitude, c.phase, c.frequency, c.offset
    a["node_count"] = count
    return a


def compile_tree_arrays(arrays, lanes, n_threads=0, count_only=False):
    """rem2d_compile_trees on node tables -> Morphology with `lanes` lanes per creature (count_only: the body counts, int32 [n],
    nothing else computed -- the static cost key of evaluate.shard_balanced)."""
    import ctypes as C
    from . import _lib
    from .compiler import TERRAIN_HEIGHT
    n, M = arrays["index"].shape
    m = Morphology(0 if count_only else n, lanes)
    B = _lib.TreeBatch()
    B.n, B.max_nodes = int(n), int(M)
    keep = []
    for k in ("node_count",) + _TREE_I32 + _TREE_F64:
        v = np.ascontiguousarray(arrays[k])
        keep.append(v)
        setattr(B, k, v.ctypes.data_as(C.c_void_p))
    out = _lib.Morph()
    for k in m.arrays:
        setattr(out, k, m.arrays[k].ctypes.data_as(C.c_void_p))
    nb = np.zeros(n, dtype=np.int32) if count_only else m.n_bodies
    _lib.check(_lib.lib().rem2d_compile_trees(C.byref(B), float(TERRAIN_HEIGHT), 64 if count_only else int(lanes),
                                              None if count_only else C.byref(out), nb.ctypes.data_as(C.c_void_p),
                                              int(n_threads)))
    return nb if count_only else m


def encode_trees_native(individuals, tree_depth=None, n_threads=0):
    """Same result as encode_population for any encoding: ``genome.create()`` in Python (for the direct encoding that
    is handing out the tree the genome already is), everything after it -- create_robot, connection sites, joint
    anchors, island order, schedule, SoA packing -- natively on all host threads."""
    if not individuals:
        return []
    trees, mls = [], []
    for ind in individuals:
        trees.append(ind.genome.create(tree_depth if tree_depth is not None else ind.tree_depth))
        mls.append(ind.genome.moduleList)
    arrays = tree_batch_arrays(trees, mls)
    M = arrays["index"].shape[1]
    lanes = 64 if M > 32 else lanes_for(M)
    return batches_from_compiled(compile_tree_arrays(arrays, lanes, n_threads))


# ------------------------------------------------------------------------------------------------
# native path for network genomes: the NN queries that grow the tree run in rem2d_compile_network too
# ------------------------------------------------------------------------------------------------
_PROTO_F64 = ("mod_width", "mod_height", "mod_radius", "mod_angle", "mod_torque", "ctl_amp", "ctl_phase", "ctl_freq",
              "ctl_offset")


def network_genome_arrays(genomes):
    """SoA view of a list of ``NNEncoding`` genomes with ``FeedForwardCPPN`` networks: weights, activation ids and the
    (mutated) module prototypes each genome carries."""
    n = len(genomes)
    T = len(genomes[0].moduleList) if n else 0
    H = genomes[0].nn_g.n_hidden if n else 0
    a = dict(w1=np.zeros((n, H, 4)), a1=np.zeros((n, H), np.int32), w2=np.zeros((n, 10, H + 1)),
             mod_shape=np.zeros((n, T), np.int32))
    for k in _PROTO_F64:
        a[k] = np.zeros((n, T), np.float64)
    for e, g in enumerate(genomes):
        net = g.nn_g
        if len(g.moduleList) != T or net.n_hidden != H or net.n_inputs != 3 or net.n_outputs != 10:
            raise ValueError("all genomes of one batch must share module count and network shape (3 -> H -> 10)")
        a["w1"][e], a["a1"][e], a["w2"][e] = net.w1, net.a1, net.w2
        for t, m in enumerate(g.moduleList):
            box = m.type == "SIMPLE"
            a["mod_shape"][e, t] = 1 if box else 2
            if box:
                a["mod_width"][e, t], a["mod_height"][e, t] = m.width, m.height
            else:
                a["mod_radius"][e, t] = m.radius
            a["mod_angle"][e, t], a["mod_torque"][e, t] = m.angle, m.torque
            c = m.controller
            a["ctl_amp"][e, t], a["ctl_phase"][e, t], a["ctl_freq"][e, t], a["ctl_offset"][e, t] = \
                c.amplitude, c.phase, c.frequency, c.offset
    return a


def compile_network_arrays(arrays, tree_depth, max_modules, lanes, n_threads=0, count_only=False):
    """rem2d_compile_network on genome arrays -> Morphology with `lanes` lanes per creature (count_only: the body counts only)."""
    import ctypes as C
    from . import _lib
    from .compiler import TERRAIN_HEIGHT
    from .controller import Controller
    from .modules import Circular2D, Standard2D
    n, T = arrays["mod_shape"].shape
    m = Morphology(0 if count_only else n, lanes)
    G = _lib.NetworkGenomes()
    G.n, G.n_types, G.n_hidden, G.max_modules = int(n), int(T), int(arrays["a1"].shape[1]), int(max_modules)
    keep = []
    for k in ("w1", "a1", "w2", "mod_shape") + _PROTO_F64:
        v = np.ascontiguousarray(arrays[k])
        keep.append(v)
        setattr(G, k, v.ctypes.data_as(C.c_void_p))
    # the classes' constants are handed over, not restated in C++
    G.box_min_width, G.box_max_width = Standard2D.MIN_WIDTH, Standard2D.MAX_WIDTH
    G.box_min_height, G.box_max_height = Standard2D.MIN_HEIGHT, Standard2D.MAX_HEIGHT
    G.box_min_angle, G.box_max_angle = Standard2D.MIN_ANGLE, Standard2D.MAX_ANGLE
    G.circle_min_radius, G.circle_max_radius = Circular2D.MIN_RADIUS, Circular2D.MAX_RADIUS
    G.circle_min_angle, G.circle_max_angle = Circular2D.MIN_ANGLE, Circular2D.MAX_ANGLE
    G.ctl_max_amp, G.ctl_max_phase = Controller.MAX_AMP, Controller.MAX_PHASE
    G.ctl_max_offset, G.ctl_max_freq = Controller.MAX_OFFSET, Controller.MAX_FREQ
    out = _lib.Morph()
    for k in m.arrays:
        setattr(out, k, m.arrays[k].ctypes.data_as(C.c_void_p))
    nb = np.zeros(n, dtype=np.int32) if count_only else m.n_bodies
    _lib.check(_lib.lib().rem2d_compile_network(C.byref(G), int(tree_depth), float(TERRAIN_HEIGHT),
                                                64 if count_only else int(lanes), None if count_only else C.byref(out),
                                                nb.ctypes.data_as(C.c_void_p), int(n_threads)))
    return nb if count_only else m


def encode_network_native(individuals, tree_depth=None, n_threads=0):
    """Same result as encode_population for ``NNEncoding`` genomes (feed-forward CPPN): tree growth by network queries
    (Network_Encoding.py:86-139,171-206), create_robot, schedule and packing all natively."""
    genomes = [ind.genome for ind in individuals]
    if not genomes:
        return []
    depth = tree_depth if tree_depth is not None else individuals[0].tree_depth
    maxm = genomes[0].maxModules
    if any(g.maxModules != maxm for g in genomes):
        raise ValueError("genomes of one batch must share maxModules")
    arrays = network_genome_arrays(genomes)
    lanes = 32   # the emitter stops at MAX_MODULES = 20 (+ the root): at most 22 nodes
    return batches_from_compiled(compile_network_arrays(arrays, depth, maxm, lanes, n_threads))
