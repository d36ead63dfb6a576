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
        idx = np.asarray(idx, dtype=np.int64)
        return LSystemPopulation({k: v[idx].copy() for k, v in self.a.items()}, self.tree_depth, self.max_modules)

    def _mutate_modules(self, morph_rate, rate, sigma, rng):
        a = self.a
        n, T = a["mod_shape"].shape
        box = a["mod_shape"] == BOX

        def jitter(x, mask, s):
            hit = mask & (rng.random((n, T)) < morph_rate)
            return np.where(hit, rng.normal(x, s), x)
        a["mod_width"] = jitter(a["mod_width"], box, sigma)
        a["mod_height"] = jitter(a["mod_height"], box, sigma)
        a["mod_radius"] = jitter(a["mod_radius"], ~box, sigma)
        a["mod_angle"] = jitter(a["mod_angle"], np.ones_like(box), sigma * math.pi)
        # limitWH
        a["mod_width"] = np.where(box, np.clip(a["mod_width"], Standard2D.MIN_WIDTH, Standard2D.MAX_WIDTH), 0.0)
        a["mod_height"] = np.where(box, np.clip(a["mod_height"], Standard2D.MIN_HEIGHT, Standard2D.MAX_HEIGHT), 0.0)
        a["mod_radius"] = np.where(box, 0.0, np.clip(a["mod_radius"], Circular2D.MIN_RADIUS, Circular2D.MAX_RADIUS))
        a["mod_angle"] = np.where(box, np.clip(a["mod_angle"], Standard2D.MIN_ANGLE, Standard2D.MAX_ANGLE),
                                  np.clip(a["mod_angle"], Circular2D.MIN_ANGLE, Circular2D.MAX_ANGLE))
        # Controller.mutate: value += gauss(value, sigma)  (m_controller.py:51-58), then minMax(angle)
        for key, s in (("ctl_amp", sigma), ("ctl_phase", sigma), ("ctl_freq", sigma * 0.1), ("ctl_offset", sigma)):
            hit = rng.random((n, T)) < rate
            a[key] = np.where(hit, a[key] + rng.normal(a[key], s), a[key])
        a["ctl_amp"] = np.clip(a["ctl_amp"], 0, Controller.MAX_AMP)
        a["ctl_phase"] = np.clip(a["ctl_phase"], -Controller.MAX_PHASE, Controller.MAX_PHASE)
        a["ctl_freq"] = np.clip(a["ctl_freq"], -Controller.MAX_FREQ, Controller.MAX_FREQ)
        half = a["mod_angle"] / 2
        a["ctl_offset"] = np.minimum(np.maximum(a["ctl_offset"], -half), half)

    def mutate(self, morph_rate, rate, sigma, rng):
        """LSystem.mutate (LSystem.py:174-179): every module, then every rule (whose mutate starts with its
        module once more, LSystem.py:71), rule growth / shrinkage with probability morph_rate each."""
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
