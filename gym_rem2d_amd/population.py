"""Array-based L-system populations: the whole EA tier without per-individual Python objects.

For a 1 M-individual generation (BASELINE config 5) the object genomes of ``encodings/lsystem.py`` cost
0.3 ms to clone + mutate and 0.5 ms to express -- minutes per generation next to seconds on the GPUs.  Here a
population is the SoA the native compiler (``rem2d_compile_lsystem``) reads: selection is a fancy index,
mutation is vectorised numpy, expression is native code (3 us per individual).

The operators are the reference's (``Encodings/LSystem.py:70-92,174-179``, ``simple_module.py:70-85``,
``circular_module.py:67-80``, ``Controller/m_controller.py:50-58`` incl. its ``x += gauss(x, sigma)`` quirk and
the double application of the module mutation per rule), applied element-wise with a numpy generator: the
same distributions, not the same random stream as ``random.*`` in the reference (the object path in
``encodings/`` keeps the exact stream and is pinned against the reference's fixtures; ``to_genomes`` /
``from_genomes`` convert between the two, and both express to identical creatures).
"""
import math

import numpy as np

from . import encode
from .controller import Controller
from .modules import Circular2D, Standard2D

BOX, CIRCLE = 1, 2


def mutate_module_arrays(a, morph_rate, rate, sigma, rng, prefix="mod_", valid=None):
    """``module.mutate(morph_rate, rate, sigma)`` (simple_module.py:70-85, circular_module.py:67-80) followed by
    ``controller.mutate(rate, sigma, module.angle)`` (m_controller.py:50-58, incl. its ``x += gauss(x, sigma)`` quirk), element-wise
    on the arrays a[prefix + "shape" | "width" | "height" | "radius" | "angle"] and a["ctl_amp" | "ctl_phase" | "ctl_freq" |
    "ctl_offset"] of one shape.  valid: bool mask of the entries that exist (padding stays untouched); None: all."""
    shape = a[prefix + "shape"]
    box = shape == BOX
    every = np.ones(shape.shape, bool) if valid is None else valid

    def jitter(x, mask, s):   # x = gauss(x, s) where the draw hits (normals are drawn for the hits only)
        idx = np.flatnonzero(mask & every & (rng.random(shape.shape) < morph_rate))
        x = np.array(x, dtype=np.float64)
        x.flat[idx] = rng.normal(x.flat[idx], s)
        return x
    W, H, R, A = prefix + "width", prefix + "height", prefix + "radius", prefix + "angle"
    a[W] = jitter(a[W], box, sigma)
    a[H] = jitter(a[H], box, sigma)
    a[R] = jitter(a[R], ~box, sigma)
    a[A] = jitter(a[A], every, sigma * math.pi)
    # limitWH (part of mutate: only what was visited is clamped -- a node born with the un-clamped default width keeps it
    # until its first visit, simple_module.py:41-42)
    a[W] = np.where(box & every, np.clip(a[W], Standard2D.MIN_WIDTH, Standard2D.MAX_WIDTH), a[W])
    a[H] = np.where(box & every, np.clip(a[H], Standard2D.MIN_HEIGHT, Standard2D.MAX_HEIGHT), a[H])
    a[R] = np.where(~box & every, np.clip(a[R], Circular2D.MIN_RADIUS, Circular2D.MAX_RADIUS), a[R])
    a[A] = np.where(every, np.where(box, np.clip(a[A], Standard2D.MIN_ANGLE, Standard2D.MAX_ANGLE),
                                    np.clip(a[A], Circular2D.MIN_ANGLE, Circular2D.MAX_ANGLE)), a[A])
    # Controller.mutate: value += gauss(value, sigma), then minMax(angle)
    for key, sg in (("ctl_amp", sigma), ("ctl_phase", sigma), ("ctl_freq", sigma * 0.1), ("ctl_offset", sigma)):
        idx = np.flatnonzero(every & (rng.random(shape.shape) < rate))
        x = np.array(a[key], dtype=np.float64)
        x.flat[idx] = x.flat[idx] + rng.normal(x.flat[idx], sg)
        a[key] = x
    a["ctl_amp"] = np.where(every, np.clip(a["ctl_amp"], 0, Controller.MAX_AMP), a["ctl_amp"])
    a["ctl_phase"] = np.where(every, np.clip(a["ctl_phase"], -Controller.MAX_PHASE, Controller.MAX_PHASE), a["ctl_phase"])
    a["ctl_freq"] = np.where(every, np.clip(a["ctl_freq"], -Controller.MAX_FREQ, Controller.MAX_FREQ), a["ctl_freq"])
    half = a[A] / 2
    a["ctl_offset"] = np.where(every, np.minimum(np.maximum(a["ctl_offset"], -half), half), a["ctl_offset"])


def random_controller_arrays(shape, rng):
    """``Controller()`` (m_controller.py:9-15) for every entry: uniform amplitude, phase, frequency, offset."""
    return dict(ctl_amp=rng.uniform(0, Controller.MAX_AMP, shape),
                ctl_phase=rng.uniform(-Controller.MAX_PHASE, Controller.MAX_PHASE, shape),
                ctl_freq=rng.uniform(-Controller.MAX_FREQ, Controller.MAX_FREQ, shape),
                ctl_offset=rng.uniform(-Controller.MAX_OFFSET, Controller.MAX_OFFSET, shape))


def default_module_arrays(n, n_box, n_circle, prefix="mod_"):
    """``get_module_list()`` (REM2D_main.py:69-77) as arrays [n][n_box + n_circle]: default module sizes (simple_module.py:41-43,
    circular_module.py:44-46)."""
    T = n_box + n_circle
    shape = np.empty((n, T), np.int32)
    shape[:, :n_box], shape[:, n_box:] = BOX, CIRCLE
    box = shape == BOX
    return {prefix + "shape": shape, prefix + "width": np.where(box, 0.2, 0.0), prefix + "height": np.where(box, 0.8, 0.0),
            prefix + "radius": np.where(box, 0.0, 0.25), prefix + "angle": np.full((n, T), math.pi / 2),
            prefix + "torque": np.full((n, T), 50.0)}


class LSystemPopulation:
    def __init__(self, arrays, tree_depth=8, max_modules=20):
        self.a = arrays
        self.tree_depth, self.max_modules = int(tree_depth), int(max_modules)

    def __len__(self):
        return int(self.a["mod_shape"].shape[0])

    # ------------------------------------------------------------------ construction
    @classmethod
    def random(cls, n, rng, n_box=4, n_circle=4, tree_depth=8, max_modules=20):
        """``Individual.random`` for n individuals: default module sizes (simple_module.py:41-43,
        circular_module.py:44-46), uniform controllers (m_controller.py:9-15), random rules (LSystem.py:30-45)."""
        T = n_box + n_circle
        a = dict(mod_shape=np.empty((n, T), np.int32))
        a["mod_shape"][:, :n_box], a["mod_shape"][:, n_box:] = BOX, CIRCLE
        box = a["mod_shape"] == BOX
        a["mod_width"] = np.where(box, 0.2, 0.0)
        a["mod_height"] = np.where(box, 0.8, 0.0)
        a["mod_radius"] = np.where(box, 0.0, 0.25)
        a["mod_angle"] = np.full((n, T), math.pi / 2)
        a["mod_torque"] = np.full((n, T), 50.0)
        a["ctl_amp"] = rng.uniform(0, Controller.MAX_AMP, (n, T))
        a["ctl_phase"] = rng.uniform(-Controller.MAX_PHASE, Controller.MAX_PHASE, (n, T))
        a["ctl_freq"] = rng.uniform(-Controller.MAX_FREQ, Controller.MAX_FREQ, (n, T))
        a["ctl_offset"] = rng.uniform(-Controller.MAX_OFFSET, Controller.MAX_OFFSET, (n, T))
        max_children = np.where(box, 3, 0)
        a["rule_n"] = (rng.integers(0, 4, (n, T)) * (max_children > 0)).astype(np.int32)   # randint(0, max_children)
        # sites without replacement: a random permutation of (left, right, top) per rule
        a["rule_site"] = np.argsort(rng.random((n, T, 3)), axis=2).astype(np.int32)
        a["rule_ref"] = rng.integers(0, T, (n, T, 3)).astype(np.int32)
        return cls(a, tree_depth, max_modules)

    @classmethod
    def from_genomes(cls, genomes):
        g0 = genomes[0]
        return cls(encode.lsystem_genome_arrays(genomes), g0.treeDepth, g0.maxModules)

    def to_genomes(self):
        """Object genomes (encodings.lsystem.LSystem) with the same parameters."""
        from .encodings.lsystem import LSystem, Symbol
        from .modules import BoxConnection
        import random
        sites = [BoxConnection.left, BoxConnection.right, BoxConnection.top]
        out = []
        state = random.getstate()
        for e in range(len(self)):
            ml = []
            for t in range(self.a["mod_shape"].shape[1]):
                if self.a["mod_shape"][e, t] == BOX:
                    m = Standard2D()
                    m.width, m.height = float(self.a["mod_width"][e, t]), float(self.a["mod_height"][e, t])
                else:
                    m = Circular2D()
                    m.radius = float(self.a["mod_radius"][e, t])
                m.angle, m.torque = float(self.a["mod_angle"][e, t]), float(self.a["mod_torque"][e, t])
                c = m.controller
                c.amplitude, c.phase = float(self.a["ctl_amp"][e, t]), float(self.a["ctl_phase"][e, t])
                c.frequency, c.offset = float(self.a["ctl_freq"][e, t]), float(self.a["ctl_offset"][e, t])
                ml.append(m)
            g = LSystem(ml)
            g.treeDepth, g.maxModules = self.tree_depth, self.max_modules
            for t, r in enumerate(g.rules):
                r.module.children = []
                r.module.availableConnections = list(ml[t].available)
                r.n_children = int(self.a["rule_n"][e, t])
                for k in range(r.n_children):
                    ref = int(self.a["rule_ref"][e, t, k])
                    s = Symbol(-1, ml[ref], ref)
                    s.parentConnectionSite = sites[int(self.a["rule_site"][e, t, k])]
                    r.module.availableConnections.remove(s.parentConnectionSite)
                    r.module.children.append(s)
            out.append(g)
        random.setstate(state)
        return out

    # ------------------------------------------------------------------ variation
    def select(self, idx):
        """Clone the individuals `idx` (tools.selTournament + toolbox.clone, REM2D_main.py:283-285)."""
        return LSystemPopulation(_take(self.a, np.asarray(idx, dtype=np.int64)), self.tree_depth, self.max_modules)

    def _mutate_modules(self, morph_rate, rate, sigma, rng):
        mutate_module_arrays(self.a, morph_rate, rate, sigma, rng)

    def mutate(self, morph_rate, rate, sigma, rng):
        """LSystem.mutate (LSystem.py:174-179): every module, then every rule (whose mutate starts with its
        module once more, LSystem.py:71), rule growth / shrinkage with probability morph_rate each."""
        if len(self) >= _MUTATE_PARALLEL_FROM:
            return _mutate_in_blocks(self, lambda a: LSystemPopulation(a, self.tree_depth, self.max_modules), rng,
                                     morph_rate, rate, sigma)
        self._mutate_serial(morph_rate, rate, sigma, rng)

    def _mutate_serial(self, morph_rate, rate, sigma, rng):
        a = self.a
        n, T = a["mod_shape"].shape
        self._mutate_modules(morph_rate, rate, sigma, rng)
        self._mutate_modules(morph_rate, rate, sigma, rng)
        max_children = np.where(a["mod_shape"] == BOX, 3, 0)
        # grow: a product on a random free site (the sites not among the first rule_n entries of rule_site)
        grow = (rng.random((n, T)) < morph_rate) & (a["rule_n"] < max_children - 1)
        if grow.any():
            e, t = np.nonzero(grow)
            k = a["rule_n"][e, t]
            free = 3 - k                                        # free sites are rule_site[e, t, k:]
            pick = k + (rng.random(len(e)) * free).astype(np.int64)
            chosen = a["rule_site"][e, t, pick].copy()
            a["rule_site"][e, t, pick] = a["rule_site"][e, t, k]
            a["rule_site"][e, t, k] = chosen
            a["rule_ref"][e, t, k] = rng.integers(0, T, len(e))
            a["rule_n"][e, t] = k + 1
        # shrink: drop a random product, its site becomes free again
        shrink = (rng.random((n, T)) < morph_rate) & (a["rule_n"] > 0)
        if shrink.any():
            e, t = np.nonzero(shrink)
            k = a["rule_n"][e, t]
            victim = (rng.random(len(e)) * k).astype(np.int64)
            vs, vr = a["rule_site"][e, t, victim].copy(), a["rule_ref"][e, t, victim].copy()
            for j in range(2):                                  # close the gap, keeping the order of the rest
                move = victim + j + 1 < k
                src = np.minimum(victim + j + 1, 2)
                dst = np.minimum(victim + j, 2)
                a["rule_site"][e[move], t[move], dst[move]] = a["rule_site"][e[move], t[move], src[move]]
                a["rule_ref"][e[move], t[move], dst[move]] = a["rule_ref"][e[move], t[move], src[move]]
            a["rule_site"][e, t, k - 1] = vs
            a["rule_ref"][e, t, k - 1] = vr
            a["rule_n"][e, t] = k - 1

    # ------------------------------------------------------------------ expression
    def body_counts(self, n_threads=0):
        """Bodies of every individual's creature (native expression, nothing else computed): the static cost key of
        evaluate.shard_balanced."""
        return encode.count_lsystem_bodies(self.a, self.tree_depth, self.max_modules, n_threads)

    def compile(self, n_threads=0):
        """Native genotype -> phenotype -> per-lane-count Morphology batches (BatchedModular2D._upload input)."""
        lanes = 64 if self.max_modules + 1 > 32 else encode.lanes_for(self.max_modules + 1)
        return encode.batches_from_compiled(
            encode.compile_lsystem_arrays(self.a, self.tree_depth, self.max_modules, lanes, n_threads))


def tournament(fitness, k, rng, tournsize=4):
    """deap.tools.selTournament, vectorised: k winners of `tournsize` uniformly drawn aspirants each."""
    fitness = np.asarray(fitness)
    asp = rng.integers(0, len(fitness), (k, tournsize))
    return asp[np.arange(k), np.argmax(fitness[asp], axis=1)]


def run_generations(pop, n_generations, evaluate, rng, morph_rate=0.01, rate=0.01, sigma=0.1, log=None):
    """The generational loop of run2D.run_deap (REM2D_main.py:280-298) on arrays.
    evaluate(LSystemPopulation) -> fitness array.  Returns (population, fitness, history)."""
    fit = np.asarray(evaluate(pop), dtype=np.float64)
    history = []
    for gen in range(n_generations):
        off = pop.select(tournament(fit, len(pop), rng))
        off.mutate(morph_rate, rate, sigma, rng)
        fit = np.asarray(evaluate(off), dtype=np.float64)
        pop = off
        history.append((gen + 1, float(fit.min()), float(fit.max()), float(fit.mean())))
        if log:
            log("Generation %d : Min %s, Max %s, Avg %s" % history[-1])
    return pop, fit, history


def sharded_evaluator(local_eval, group=None, device=None, on_error="raise", balance=True, n_threads=0):
    """evaluate(population) for a torch.distributed job (one process per GPU): every rank holds the whole
    population as arrays (selection and mutation are replicated from a shared seed, so no genome ever crosses a
    rank boundary), expresses and evaluates only its contiguous block ``[lo, hi)`` with
    ``local_eval(LSystemPopulation block) -> fitness[hi-lo]`` and the ranks exchange one all-gather of float64
    fitness (REM2D_main.py:256-267 pool.map).

    balance (default): the reference's pool balances dynamically (pool.map hands out chunks as workers finish); a static
    contiguous cut does not, and a generation ends with its slowest rank.  Every rank therefore computes the same deal from the
    arrays it holds anyway -- the individuals in descending order of body count (a native expression pass, population.body_counts)
    dealt snake-wise over the ranks, evaluate.shard_balanced -- and the fitness comes back in population order through the same
    single all-gather.  ``evaluate.last_shard_cost`` = the predicted cost (bodies) per rank.  balance=False: contiguous blocks.

    on_error: what to do about creatures without a valid fitness (beyond even the wide build's contact slots -- Box2D has
    no such cap, so the reference would have produced a number).  "raise" (the library default): every rank raises
    SolverOverflow AFTER the collective, in step.  "penalty" (what an EA loop opts into so that one out-of-domain creature
    does not abort a run): evaluate.UNRESOLVED_FITNESS on every rank alike, a warning on EVERY rank, and the indices kept per
    call in ``evaluate.unresolved_log`` (a list that grows by one entry per call: a long run cannot lose them)."""
    import torch
    import torch.distributed as dist
    from .evaluate import all_gather_fitness, shard_balanced, shard_costs, shard_range

    def evaluate(pop):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        index = None
        if balance:
            cost = pop.body_counts(n_threads)
            index = shard_balanced(cost, world)
            mine = index[rank][index[rank] >= 0]
            evaluate.last_shard_cost = shard_costs(cost, index)
        else:
            lo, hi = shard_range(len(pop), rank, world)
            mine = np.arange(lo, hi)
        local = local_eval(pop.select(mine)) if len(mine) else np.zeros(0, dtype=np.float64)
        mask = None
        if isinstance(local, tuple):     # (fitness, unresolved mask): gpu_evaluator(masked=True)
            local, mask = local
        local = torch.as_tensor(np.asarray(local, dtype=np.float64), device=device)
        mask = torch.zeros(local.numel(), dtype=torch.bool) if mask is None else torch.as_tensor(np.asarray(mask))
        # the mask rides in the fitness all-gather: a rank with a creature that has no valid fitness does not raise
        # before the collective (the others would wait in it for ever).  After it every rank holds the same mask: the
        # creatures get the defined penalty on every rank alike (on_error="penalty": the generation goes on, the indices
        # are kept in evaluate.last_unresolved) or every rank raises in step (on_error="raise")
        fit, bad = all_gather_fitness(local, len(pop), group, flags=mask.to(local.device), index=index)
        evaluate.last_unresolved = []
        if bool(bad.any()):
            from .evaluate import SolverOverflow, apply_penalty
            from . import _lib
            if on_error != "penalty":
                idx = torch.nonzero(bad).flatten().cpu().tolist()
                raise SolverOverflow(idx, [_lib.ERR_SOLVER_OVERFLOW] * len(idx))
            evaluate.last_unresolved = apply_penalty(fit, bad, warn=True)
        evaluate.unresolved_log.append(list(evaluate.last_unresolved))
        return fit.cpu().numpy().astype(np.float64)
    evaluate.last_unresolved = []
    evaluate.unresolved_log = []
    evaluate.last_shard_cost = None
    return evaluate


def gpu_evaluator(env=None, max_steps=None, n_threads=0, masked=False, on_error="fallback"):
    """local_eval for sharded_evaluator / evaluate for run_generations on one GPU: native expression, upload,
    whole episodes (evaluate()'s rule), fitness as float64 numpy.  Creatures that overflow the default build's contact
    slots are re-evaluated in the wide build (evaluate.reevaluate_wide); one that overflows even that gets
    raises SolverOverflow (on_error="fallback", the library default: Box2D has no contact cap, so there is no fitness the
    reference would have given) or, if the caller opts in with on_error="penalty" (the EA loops do: one out-of-domain
    creature does not abort a generation), gets evaluate.UNRESOLVED_FITNESS, a warning, and an entry in
    ``evaluate.last_unresolved`` / ``evaluate.unresolved_log`` (one list per call) -- or, with masked=True
    (what a sharded job wants), comes back in a second array ``(fitness, unresolved)`` so that the verdict is taken after
    the job's collective."""
    from .env import BatchedModular2D
    from .evaluate import EPISODE_CAP, run_episode, run_episode_masked
    holder = {"env": env}

    def evaluate(pop):
        if holder["env"] is None:
            from . import _lib
            holder["env"] = BatchedModular2D(flags=_lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN)
        e = holder["env"]
        e.trees = e.robots = None
        e._upload(pop.compile(n_threads), len(pop))
        cap = max_steps if max_steps is not None else EPISODE_CAP
        if masked:
            fit, bad = run_episode_masked(e, cap)
            return fit.cpu().numpy(), bad.cpu().numpy()
        fit = run_episode(e, cap, on_error=on_error).cpu().numpy()
        evaluate.last_unresolved = list(getattr(e, "last_unresolved", []))
        evaluate.unresolved_log.append(list(evaluate.last_unresolved))
        return fit
    evaluate.last_unresolved = []
    evaluate.unresolved_log = []
    return evaluate


# ------------------------------------------------------------------------------------------------------------------------
# Array populations of the other two encodings (round 5).  Same contract as LSystemPopulation: the reference's operators
# element-wise with a numpy generator -- the same distributions, not the reference's random stream (the object genomes in
# encodings/ keep the exact stream and are pinned against the reference's fixtures) -- and a compile() that feeds the native
# compilers directly.
# ------------------------------------------------------------------------------------------------------------------------
_MUTATE_BLOCKS = 16          # row blocks of a big population's mutation (a constant: the result does not depend on the host's cores)
_MUTATE_PARALLEL_FROM = 32768


def _mutate_in_blocks(pop, make, rng, *args):
    """pop.mutate(*args, rng) for a big population: _MUTATE_BLOCKS contiguous row blocks, each with a generator spawned from `rng`
    (numpy.random.Generator.spawn: independent streams, reproducible from the seed), mutated side by side on a thread pool --
    numpy releases the GIL inside its loops and generators -- and written back.  make(arrays) -> a population of pop's type."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(pop)
    edges = np.linspace(0, n, _MUTATE_BLOCKS + 1).astype(np.int64)
    gens = rng.spawn(_MUTATE_BLOCKS)

    def work(b):
        lo, hi = int(edges[b]), int(edges[b + 1])
        sub = make({k: v[lo:hi].copy() for k, v in pop.a.items()})
        sub._mutate_serial(*args, gens[b])
        for k, v in sub.a.items():
            pop.a[k][lo:hi] = v
    with ThreadPoolExecutor(max_workers=min(_MUTATE_BLOCKS, len(__import__("os").sched_getaffinity(0)))) as ex:
        list(ex.map(work, range(_MUTATE_BLOCKS)))


def _take(arrays, idx):
    """arrays[k][idx] for every k, on a thread pool (a 1 M-individual population is gigabytes of fancy indexing; numpy's take
    releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    keys = list(arrays)
    if len(idx) < 65536:
        return {k: arrays[k][idx] for k in keys}
    with ThreadPoolExecutor(max_workers=min(len(keys), 16)) as ex:
        return dict(zip(keys, ex.map(lambda k: np.take(arrays[k], idx, axis=0), keys)))


class NetworkPopulation:
    """Network-encoded individuals (Encodings/Network_Encoding.py:42-222 with this build's feed-forward CPPN genome,
    encodings/network.py: 3 -> n_hidden -> 10) as arrays: weights w1 [n][H][4], activation ids a1 [n][H], w2 [n][10][H + 1] and
    the T module prototypes every genome carries and mutates (NN_enc.moduleList)."""

    def __init__(self, arrays, tree_depth=7, max_modules=20):
        self.a = arrays
        self.tree_depth, self.max_modules = int(tree_depth), int(max_modules)

    def __len__(self):
        return int(self.a["a1"].shape[0])

    @classmethod
    def random(cls, n, rng, n_box=4, n_circle=4, n_hidden=8, tree_depth=7, max_modules=20):
        """``Individual.random(encoding='cppn')``: FeedForwardCPPN() weights (gauss(0, 1.5) / gauss(0, 1), a random activation per
        hidden node), the default module list with fresh controllers, every prototype mutated once with (0.5, 0.5, 0.5)
        (NNEncoding.__init__ / Network_Encoding.py:80-84)."""
        a = default_module_arrays(n, n_box, n_circle)
        a.update(random_controller_arrays(a["mod_shape"].shape, rng))
        a["w1"] = rng.normal(0.0, 1.5, (n, n_hidden, 4))
        a["a1"] = rng.integers(0, 4, (n, n_hidden)).astype(np.int32)
        a["w2"] = rng.normal(0.0, 1.0, (n, 10, n_hidden + 1))
        mutate_module_arrays(a, 0.5, 0.5, 0.5, rng)
        return cls(a, tree_depth, max_modules)

    @classmethod
    def from_genomes(cls, genomes, tree_depth=7):
        return cls(encode.network_genome_arrays(genomes), tree_depth, genomes[0].maxModules)

    def select(self, idx):
        return NetworkPopulation(_take(self.a, np.asarray(idx, dtype=np.int64)), self.tree_depth, self.max_modules)

    def mutate(self, morph_rate, rate, sigma, rng, weight_rate=0.2, weight_sigma=0.3):
        """NNEncoding.mutate (Network_Encoding.py:142-152): the network's own mutation (FeedForwardCPPN.mutate: every weight
        with probability 0.2 += gauss(0, 0.3)), then every module prototype."""
        if len(self) >= _MUTATE_PARALLEL_FROM:
            return _mutate_in_blocks(self, lambda a: NetworkPopulation(a, self.tree_depth, self.max_modules), rng,
                                     morph_rate, rate, sigma, weight_rate, weight_sigma)
        self._mutate_serial(morph_rate, rate, sigma, weight_rate, weight_sigma, rng)

    def _mutate_serial(self, morph_rate, rate, sigma, weight_rate, weight_sigma, rng):
        for k in ("w1", "w2"):
            w = self.a[k]
            hit = rng.random(w.shape) < weight_rate
            w[hit] += rng.normal(0.0, weight_sigma, int(hit.sum()))
        mutate_module_arrays(self.a, morph_rate, rate, sigma, rng)

    def body_counts(self, n_threads=0):
        return encode.compile_network_arrays(self.a, self.tree_depth, self.max_modules, 32, n_threads, count_only=True)

    def compile(self, n_threads=0):
        """Native growth of the trees by network queries, create_robot, schedule, packing (rem2d_compile_network)."""
        return encode.batches_from_compiled(
            encode.compile_network_arrays(self.a, self.tree_depth, self.max_modules, 32, n_threads))


def _PROTO_DEFAULTS():
    """(box width, box height, circle radius, angle, torque) of a freshly constructed module (simple_module.py:41-52,
    circular_module.py:44-52): what a new node of the direct encoding copies from the never-mutated prototype list."""
    import random
    state = random.getstate()       # (the constructors draw their controllers from `random`: leave the stream alone)
    b, c = Standard2D(), Circular2D()
    random.setstate(state)
    assert b.angle == c.angle and b.torque == c.torque
    return float(b.width), float(b.height), float(c.radius), float(b.angle), float(b.torque)


_SITES = 3   # BoxConnection: 0 left, 1 right, 2 top (circles have no sites)
_NODE_F64 = ("width", "height", "radius", "angle", "torque", "ctl_amp", "ctl_phase", "ctl_freq", "ctl_offset")


class DirectPopulation:
    """Direct-encoding individuals (Encodings/Direct_Encoding.py:7-139: the genome IS the phenotype tree, one module object and
    one controller per node) as node tables [n][M] in Tree.getNodes() order -- a node's subtree is the run of columns behind it
    -- with the columns rem2d_compile_trees reads: node_count [n]; parent (column of the parent, -1 root), site (0 left, 1 right,
    2 top, -1 root), shape (1 box, 2 circle, 0 padding), width / height / radius / angle / torque and the node controller.
    M = max_modules; prototypes for new nodes: the default module list (4 boxes + 4 circles), never mutated themselves."""

    def __init__(self, arrays, max_modules=20, max_depth=8, n_box=4, n_circle=4):
        self.a = arrays
        self.max_modules, self.max_depth, self.n_box, self.n_circle = int(max_modules), int(max_depth), int(n_box), int(n_circle)

    def __len__(self):
        return int(self.a["node_count"].shape[0])

    @staticmethod
    def _empty(n, M):
        a = {k: np.zeros((n, M), np.float64) for k in _NODE_F64}
        a.update(parent=np.full((n, M), -1, np.int32), site=np.full((n, M), -1, np.int32), shape=np.zeros((n, M), np.int32),
                 node_count=np.zeros(n, np.int32))
        return a

    @classmethod
    def random(cls, n, rng, max_modules=20, max_depth=8, n_box=4, n_circle=4, numpy_only=False):
        """``DirectEncoding(moduleList)`` (Direct_Encoding.py:44-60): a root box with a fresh controller, then five rounds of
        mutate(0.5, 0.5, 0.5)."""
        bw, bh, _, ang, tor = _PROTO_DEFAULTS()
        a = cls._empty(n, max_modules)
        a["node_count"][:] = 1
        a["shape"][:, 0], a["width"][:, 0], a["height"][:, 0] = BOX, bw, bh
        a["angle"][:, 0], a["torque"][:, 0] = ang, tor
        for k, v in random_controller_arrays((n,), rng).items():
            a[k][:, 0] = v
        p = cls(a, max_modules, max_depth, n_box, n_circle)
        for _ in range(5):
            (p.mutate_numpy if numpy_only else p.mutate)(0.5, 0.5, 0.5, rng)
        return p

    @classmethod
    def from_genomes(cls, genomes):
        """From encodings.direct.DirectEncoding objects (their trees, flattened)."""
        g0 = genomes[0]
        trees = [g.create(g.maxDepth) for g in genomes]
        t = encode.tree_batch_arrays(trees, [g.moduleList for g in genomes], max_nodes=g0.maxModules)
        a = {k: t[k] for k in _NODE_F64 + ("site", "shape", "node_count")}
        # tree_batch_arrays keeps node.parent = the parent's node.index; after reassignIndices that IS its column
        a["parent"] = t["parent"].astype(np.int32)
        a["parent"][:, 0] = -1
        pad = np.arange(a["shape"].shape[1])[None, :] >= a["node_count"][:, None]
        a["parent"][pad] = -1
        return cls(a, g0.maxModules, g0.maxDepth)

    def select(self, idx):
        return DirectPopulation(_take(self.a, np.asarray(idx, dtype=np.int64)), self.max_modules, self.max_depth,
                                self.n_box, self.n_circle)

    # ---------------------------------------------------------------- structure helpers (vectorised over individuals)
    def depths(self):
        """Depth of every node (root 0), padding -1.  Parents come before their children, so one sweep over the columns."""
        a = self.a
        n, M = a["shape"].shape
        rows = np.arange(n)
        d = np.full((n, M), -1, np.int32)
        d[:, 0] = 0
        for i in range(1, M):
            live = i < a["node_count"]
            d[:, i] = np.where(live, d[rows, np.maximum(a["parent"][:, i], 0)] + 1, -1)
        return d

    def subtree_sizes(self):
        a = self.a
        n, M = a["shape"].shape
        rows = np.arange(n)
        size = (np.arange(M)[None, :] < a["node_count"][:, None]).astype(np.int32)
        for i in range(M - 1, 0, -1):
            live = i < a["node_count"]
            np.add.at(size, (rows[live], a["parent"][live, i]), size[live, i])
        return size

    def check(self):
        """Structural invariants of a node table (tests): Tree.getNodes() order, unique sites per parent, size / depth caps."""
        a = self.a
        n, M = a["shape"].shape
        cnt = a["node_count"]
        assert (cnt >= 1).all() and (cnt <= self.max_modules).all()
        live = np.arange(M)[None, :] < cnt[:, None]
        assert ((a["shape"] > 0) == live).all() and (a["parent"][:, 0] == -1).all() and (a["shape"][:, 0] == BOX).all()
        assert (a["parent"][live][1:] < np.broadcast_to(np.arange(M), (n, M))[live][1:]).all()
        d = self.depths()
        assert (d[live] <= self.max_depth).all()
        size = self.subtree_sizes()
        rows = np.arange(n)
        seen = np.zeros((n, M, _SITES), bool)
        for i in range(1, M):
            lv = i < cnt
            p, s = a["parent"][lv, i], a["site"][lv, i]
            assert (p >= 0).all() and (s >= 0).all() and (s < _SITES).all() and (a["shape"][rows[lv], p] == BOX).all()
            assert not seen[rows[lv], p, s].any()
            seen[rows[lv], p, s] = True
            # pre-order: node i lies inside its parent's run of columns
            assert (i < p + size[rows[lv], p]).all()
        return True

    # ---------------------------------------------------------------- variation
    def mutate(self, morph_rate, rate, sigma, rng, n_threads=0):
        """DirectEncoding.mutate (Direct_Encoding.py:82-139) on every individual, in place, natively (rem2d_mutate_trees: the
        reference's algorithm statement by statement, one generator per individual seeded from `rng` -- the result does not
        depend on the thread count).  ~0.1 s per million individuals; mutate_numpy is the same operator in numpy."""
        import ctypes as C
        from . import _lib
        a = self.a
        n, M = a["shape"].shape
        P = _lib.TreePopulation()
        P.n, P.max_nodes = int(n), int(M)
        for k in ("node_count", "parent", "site", "shape") + _NODE_F64:
            want = np.int32 if k in ("node_count", "parent", "site", "shape") else np.float64
            if a[k].dtype != want or not a[k].flags["C_CONTIGUOUS"]:
                a[k] = np.ascontiguousarray(a[k], dtype=want)
            setattr(P, k, a[k].ctypes.data_as(C.c_void_p))
        P.max_modules, P.max_depth, P.n_box, P.n_circle = self.max_modules, self.max_depth, self.n_box, self.n_circle
        (P.proto_box_width, P.proto_box_height, P.proto_circle_radius, P.proto_angle, P.proto_torque) = _PROTO_DEFAULTS()
        P.box_min_width, P.box_max_width = Standard2D.MIN_WIDTH, Standard2D.MAX_WIDTH
        P.box_min_height, P.box_max_height = Standard2D.MIN_HEIGHT, Standard2D.MAX_HEIGHT
        P.box_min_angle, P.box_max_angle = Standard2D.MIN_ANGLE, Standard2D.MAX_ANGLE
        P.circle_min_radius, P.circle_max_radius = Circular2D.MIN_RADIUS, Circular2D.MAX_RADIUS
        P.circle_min_angle, P.circle_max_angle = Circular2D.MIN_ANGLE, Circular2D.MAX_ANGLE
        P.ctl_max_amp, P.ctl_max_phase = Controller.MAX_AMP, Controller.MAX_PHASE
        P.ctl_max_offset, P.ctl_max_freq = Controller.MAX_OFFSET, Controller.MAX_FREQ
        seed = int(rng.integers(0, 2 ** 63 - 1))
        _lib.check(_lib.lib().rem2d_mutate_trees(C.byref(P), float(morph_rate), float(rate), float(sigma), seed, int(n_threads)))

    def mutate_numpy(self, morph_rate, rate, sigma, rng):
        """DirectEncoding.mutate (Direct_Encoding.py:82-139) on every individual at once, event by event in the reference's order
        with the individuals side by side (loops over the <= 20 columns, vector operations over the population):

        * walking down (``for mod in node.children``): every child draws against morph / 2 / n_modules.  A hit removes the child
          and its subtree -- unless the node is the root (``if depth != 0``), whose child then merely is not descended into --
          and the sibling that slides into the freed list slot is skipped for this call (python's list iterator); no hit: the
          walk descends, i.e. the child's subtree will be visited.  n_modules is recounted after every removal.
        * walking back up (``for con in node.availableConnections``): every free site of a visited box draws against
          morph / n_modules while n_modules < maxModules and depth < maxDepth; a hit appends a copy of a random prototype with a
          fresh controller (addChild: behind the node's existing children), n_modules grows by one, and the site that slides
          into the used one's list slot is skipped.  A node born in this call is not visited.
        * every visited node then mutates its module and its controller.

        Left out, in distribution only: the order of re-appended sites in ``availableConnections`` (here always left, right, top)
        and the interleaving of one subtree's removals with an earlier subtree's growth in the running count (all removals of a
        call are counted before its first growth).  Kept as an independently written second form of the operator (the tests
        hold both against the object genome); the product path is mutate()."""
        a = self.a
        n, M = a["shape"].shape
        rows = np.arange(n)
        cnt0 = a["node_count"].astype(np.int64)
        live = np.arange(M)[None, :] < cnt0[:, None]
        depth = self.depths()
        size = self.subtree_sizes()
        par = np.maximum(a["parent"], 0)
        # ---- the walk down: removals, skipped siblings, which nodes are visited
        cur = cnt0.copy()                                    # n_modules, recounted as the reference does
        gone = np.zeros((n, M), bool)
        visited = np.zeros((n, M), bool)
        visited[:, 0] = True
        skip_next = np.zeros((n, M), bool)                   # per parent: its next child is skipped
        u = rng.random((n, M))
        for i in range(1, M):
            lv = live[:, i]
            pi = par[:, i]
            pg, pv = gone[rows, pi], visited[rows, pi]
            sk = skip_next[rows, pi] & lv & pv & ~pg
            skip_next[rows[sk], pi[sk]] = False
            draw = lv & pv & ~pg & ~sk
            hit = draw & (u[:, i] < morph_rate / 2.0 / cur)
            own = hit & (depth[rows, pi] != 0)               # removed by its own draw
            gone[:, i] = lv & (pg | own)
            visited[:, i] = draw & ~hit
            cur = cur - np.where(own, size[:, i], 0)
            skip_next[rows[own], pi[own]] = True
        keep = live & ~gone
        # ---- the walk back up: growth on the free sites of the visited boxes, in post-order (a node after all its descendants)
        used = np.zeros((n, M, _SITES), bool)
        for i in range(1, M):
            k = keep[:, i]
            used[rows[k], a["parent"][k, i], a["site"][k, i]] = True
        last = np.where(keep, np.arange(M)[None, :], -1)
        for i in range(M - 1, 0, -1):
            k = keep[:, i]
            np.maximum.at(last, (rows[k], a["parent"][k, i]), last[k, i])
        grow = keep & visited & (a["shape"] == BOX) & (depth < self.max_depth)
        post = np.argsort(np.where(grow, last * 64 + (63 - depth), 1 << 20), axis=1, kind="stable")
        want = np.zeros((n, M, _SITES), bool)
        v = rng.random((n, M, _SITES))
        for k in range(int(grow.sum(axis=1).max()) if n else 0):
            j = post[:, k]
            act = grow[rows, j]
            skip = np.zeros(n, bool)
            for sidx in range(_SITES):
                free = act & ~used[rows, j, sidx]
                take = free & ~skip
                skip = skip & ~free                          # (the skipped one was this free site)
                add = take & (cur < self.max_modules) & (v[rows, j, sidx] < morph_rate / cur)
                want[rows[add], j[add], sidx] = True
                cur = cur + add
                skip = skip | add
        touched = gone.any(axis=1) | want.any(axis=(1, 2))
        old = live & visited & ~gone
        if touched.any():
            sel = np.nonzero(touched)[0]
            old[sel] = self._restructure(sel, keep, want, depth, old, rng)
        # ---- the visited nodes mutate (module, then controller)
        mutate_module_arrays(a, morph_rate, rate, sigma, rng, prefix="", valid=old)

    def _restructure(self, sel, keep, want, depth, visited_old, rng):
        """Rows `sel`: drop the removed nodes, insert the new children behind their parent's subtree (Tree.getNodes() order:
        addChild appends to node.children, and a node's own growth comes after its children's), renumber the parents.  Returns
        the mask `visited_old` (old columns) carried over to the new columns, False for the nodes born here."""
        a = self.a
        M = a["shape"].shape[1]
        m = len(sel)
        r = np.arange(m)
        keep_s, want_s, depth_s = keep[sel], want[sel], depth[sel]
        par = a["parent"][sel]
        # last column of every kept node's subtree among the kept nodes = max kept descendant column (sweep from the back)
        last = np.where(keep_s, np.arange(M)[None, :], -1)
        for i in range(M - 1, 0, -1):
            k = keep_s[:, i]
            np.maximum.at(last, (r[k], par[k, i]), last[k, i])
        # sort key of every candidate: kept old nodes keep their column; a new child of parent j at site s goes behind last[j],
        # deeper parents first (their growth precedes their ancestors'), then by site
        W = M * (1 + _SITES)
        key = np.full((m, W), np.inf)
        key[:, :M] = np.where(keep_s, np.arange(M)[None, :].astype(np.float64), np.inf)
        new_key = last[:, :, None] + 0.5 - depth_s[:, :, None] / 256.0 + np.arange(_SITES)[None, None, :] / 4096.0
        key[:, M:] = np.where(want_s, new_key, np.inf).reshape(m, M * _SITES)
        # cut the additions at max_modules in tree order: of the new candidates only the first (max_modules - kept) survive
        order = np.argsort(key, axis=1, kind="stable")
        skey = np.take_along_axis(key, order, axis=1)
        is_new = order >= M
        room = (self.max_modules - keep_s.sum(axis=1))[:, None]
        drop = is_new & np.isfinite(skey) & (np.cumsum(is_new & np.isfinite(skey), axis=1) > room)
        skey = np.where(drop, np.inf, skey)
        order2 = np.argsort(skey, axis=1, kind="stable")
        order = np.take_along_axis(order, order2, axis=1)[:, :M]
        skey = np.take_along_axis(skey, order2, axis=1)[:, :M]
        valid = np.isfinite(skey)
        cnt = valid.sum(axis=1).astype(np.int32)
        src_old = np.where(order < M, order, 0)                     # column of the old node (for the old ones)
        new_parent_old = np.where(order >= M, (order - M) // _SITES, 0)   # OLD column of a new node's parent
        new_site = np.where(order >= M, (order - M) % _SITES, -1)
        isnew = valid & (order >= M)
        # old column -> new column
        newcol = np.full((m, M), -1, np.int64)
        rr = np.broadcast_to(r[:, None], (m, M))
        oldmask = valid & (order < M)
        newcol[rr[oldmask], order[oldmask]] = np.broadcast_to(np.arange(M)[None, :], (m, M))[oldmask]
        out = self._empty(m, M)
        for k in _NODE_F64 + ("shape", "site"):
            v = np.take_along_axis(a[k][sel], src_old, axis=1)
            out[k] = np.where(oldmask, v, out[k])
        op = np.take_along_axis(par, src_old, axis=1)
        out["parent"] = np.where(oldmask, np.where(op >= 0, newcol[rr, np.maximum(op, 0)], -1), -1).astype(np.int32)
        out["parent"] = np.where(isnew, newcol[rr, new_parent_old], out["parent"]).astype(np.int32)
        out["site"] = np.where(isnew, new_site, out["site"]).astype(np.int32)
        # the new nodes: a copy of a random prototype (default sizes: the prototypes never mutate) + Controller()
        k_new = int(isnew.sum())
        if k_new:
            T = self.n_box + self.n_circle
            ref = rng.integers(0, T, k_new)
            is_box = ref < self.n_box
            out["shape"][isnew] = np.where(is_box, BOX, CIRCLE)
            bw, bh, cr, ang, tor = _PROTO_DEFAULTS()
            out["width"][isnew] = np.where(is_box, bw, 0.0)
            out["height"][isnew] = np.where(is_box, bh, 0.0)
            out["radius"][isnew] = np.where(is_box, 0.0, cr)
            out["angle"][isnew] = ang
            out["torque"][isnew] = tor
            for k, v in random_controller_arrays((k_new,), rng).items():
                out[k][isnew] = v
        out["node_count"] = cnt
        for k in out:
            a[k][sel] = out[k]
        return oldmask & np.take_along_axis(visited_old[sel], src_old, axis=1)

    # ---------------------------------------------------------------- expression
    def tree_arrays(self):
        """The rem2d_tree_batch node tables (index = column: DirectEncoding.reassignIndices)."""
        n, M = self.a["shape"].shape
        t = {k: self.a[k] for k in _NODE_F64 + ("site", "shape", "node_count", "parent")}
        t["index"] = np.broadcast_to(np.arange(M, dtype=np.int32), (n, M)).copy()
        return t

    def body_counts(self, n_threads=0):
        return encode.compile_tree_arrays(self.tree_arrays(), 32, n_threads, count_only=True)

    def compile(self, n_threads=0):
        """create_robot, connection sites, joint anchors, island order, schedule, packing: natively (rem2d_compile_trees)."""
        lanes = 64 if self.max_modules > 32 else encode.lanes_for(self.max_modules)
        return encode.batches_from_compiled(encode.compile_tree_arrays(self.tree_arrays(), lanes, n_threads))
