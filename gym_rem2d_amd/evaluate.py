"""Episode loop and population evaluation.

``evaluate`` has the reference's signature and fitness rule (``REM2D_main.py:350-378`` ==
``Demo1_Random_Individual.py:4-36``).  ``evaluate_population`` replaces
``pool.map(evaluate, population)`` (``REM2D_main.py:256-267,291``) with one batched episode on
the GPU; ``evaluate_population_sharded`` shards the population over the ranks of a
torch.distributed job (one process per GPU, contiguous blocks) and gathers the fitness with a
single all-gather -- RCCL over xGMI on the MI355X node, gloo in the CPU tests.  There is no
per-step collective: creatures are independent.
"""
import math

import numpy as np
import torch

from . import _lib

EPISODE_CAP = 2500  # wall of death advances 0.04/step and the goal is x > 100


def _env_singleton():
    global _ENV
    try:
        return _ENV
    except NameError:
        from .env import Modular2D
        _ENV = Modular2D()
        return _ENV


def evaluate(individual, EVALUATION_STEPS=10000, HEADLESS=True, INTERVAL=100, ENV_LENGTH=100, TREE_DEPTH=None,
             CONTROLLER=None):
    """One individual, one episode, reference semantics (action ignored, reward sentinels)."""
    env = _env_singleton()
    if TREE_DEPTH is None:
        try:
            TREE_DEPTH = individual.tree_depth
        except AttributeError:
            raise Exception("Tree depth not defined in evaluation")
    tree = individual.genome.create(TREE_DEPTH)
    env.seed(4)
    env.reset(tree=tree, module_list=individual.genome.moduleList)
    fitness = 0
    for i in range(EVALUATION_STEPS):
        action = np.ones_like(env.action_space.sample())
        _, reward, done, _ = env.step(action)
        if reward < -10:
            break
        elif reward > ENV_LENGTH:
            reward += (EVALUATION_STEPS - i) / EVALUATION_STEPS
            fitness = reward
            break
        if reward > 0:
            fitness = reward
    return fitness


class SolverOverflow(_lib.Rem2dError):
    """A creature ran out of engine capacity during an episode, so its trajectory is NOT what the reference's Box2D
    computes: more than REM2D_SOLVER_SLOTS touching contacts on one body (the extra manifold was left out of the solver)
    or more than REM2D_CONTACT_SLOTS broadphase pairs on one body (the extra pair was never created).
    ``indices`` are the affected population indices, ``codes`` their REM2D_ERR_* bits."""

    def __init__(self, indices, codes):
        self.indices, self.codes = indices, codes
        super().__init__("%d creature(s) overflowed the solver / pair slots (first: %s, REM2D_ERR bits %s); their fitness "
                         "is not the reference's" % (len(indices), indices[:8], sorted(set(codes))))


HandoverError = _lib.HandoverError   # (the step train's hand-over check: a launch-form failure, not a capacity one -- see _lib)

# What a creature gets whose contacts did not fit even the wide build (run_episode's on_error="penalty"): the value
# evaluate() starts from and returns for an individual that never scored (REM2D_main.py:361 ``fitness = 0``).  Box2D has
# no contact cap (Modular2DEnv.py:634), the engine here has two tiers (24 / 6, then 32 / 12 pair / solver slots per body);
# DESIGN.md 8 shows that no module the reference's classes can produce (boxes <= 1 x 1, circles r <= 0.5: limitWH,
# simple_module.py:55-68) reaches the second tier's limits, so this is the verdict on hand-built out-of-domain bodies --
# an EA generation records the mask and goes on instead of raising.
UNRESOLVED_FITNESS = 0.0


def check_errors(env, on_error="raise"):
    """Read the per-creature engine error bits of a BatchedModular2D.  REM2D_ERR_HANDOVER (the step train's hand-over check) is a
    launch-form failure and is judged first and apart: "raise" and "warn" both raise HandoverError for it (no mode scores such a
    creature), "ignore" returns it in the mask.  The capacity bits (REM2D_ERR_PAIR_OVERFLOW / _SOLVER_OVERFLOW) -- on_error:
    "raise" -> SolverOverflow with the creatures' real codes; "warn" -> warnings.warn and return the mask; "ignore" -> return the
    mask (every non-zero code)."""
    err = env.errors()
    bad = err != 0
    if on_error != "ignore" and bool(bad.any()):
        hand = (err & _lib.ERR_HANDOVER) != 0
        if bool(hand.any()):
            raise HandoverError(int(hand.sum()), " (creatures, first: %s)" % torch.nonzero(hand).flatten().cpu().tolist()[:8])
        idx = torch.nonzero(bad).flatten().cpu().tolist()
        codes = err[bad].cpu().tolist()
        if on_error == "raise":
            raise SolverOverflow(idx, codes)
        import warnings
        warnings.warn(str(SolverOverflow(idx, codes)))
    return bad


def _episode(env, max_steps, chunk, compact):
    done_steps = 0
    compact = compact and bool(env.flags & _lib.FLAG_SKIP_FROZEN)
    while done_steps < max_steps:
        n = min(chunk, max_steps - done_steps)
        env.step(n)
        done_steps += n
        if compact:
            if env.compact() == 0:
                break
        elif bool((env.frozen != 0).all()):
            break


def reevaluate(env, mask, fit, wide=None, options=None, max_steps=EPISODE_CAP, chunk=100):
    """Creatures are independent and an episode is a function of the morphology alone (fixed terrain seed, open-loop controller:
    SURVEY facts 5 / 6), so any subset can simply be evaluated AGAIN from reset in worlds of another build or launch form -- same
    kernels, same arithmetic, same bits for everything both forms can hold.  ``mask``: bool [N] in population order; wide / options:
    the build (None: the env's own) and launch options of the second run.  Writes the fitness of those creatures into ``fit`` and
    returns their REM2D_ERR_* codes of the second run as int32 [N] (0 elsewhere).  Two callers:
      * the overflow fallback (``reevaluate_wide``): Box2D has no cap on the contacts of a body (Modular2DEnv.py:634 world.Step
        solves whatever touches); the default build keeps 24 pair / 6 solver slots per body and flags the creature that needs
        more -- re-run in librem2d_wide.so (32 / 12);
      * the hand-over fallback: creatures a step train flagged REM2D_ERR_HANDOVER -- re-run in the SAME build on per-step launches
        (``options={"fuse_velpost": 1}``: no hand-over between workgroups exists there)."""
    from .env import BatchedModular2D
    idx_bad = torch.nonzero(mask).flatten().cpu().numpy()
    codes = torch.zeros(mask.shape, dtype=torch.int32, device=mask.device)
    if idx_bad.size == 0:
        return codes
    opts = dict(env.options)
    opts.pop("train_fault", None)   # (a test hook of the first run's launch form; the second run is there to be trusted)
    opts.update(options or {})
    again = BatchedModular2D(hardcore=env.hardcore, flat=env.flat, seed=env._seed, device=env.device, flags=env.flags,
                             wide=env.wide if wide is None else wide, options=opts, on_handover="flag")
    again.terrain = env._terrain()
    batches = []
    for morph, idx in env._uploaded:
        sel = np.nonzero(np.isin(idx, idx_bad))[0]
        if sel.size:
            batches.append((morph.take(sel), np.searchsorted(idx_bad, idx[sel]).tolist()))
    again._upload(batches, int(idx_bad.size))
    _episode(again, max_steps, chunk, compact=False)
    where = torch.as_tensor(idx_bad, dtype=torch.long, device=fit.device)
    fit.index_copy_(0, where, again.fitness.to(fit.device))
    codes.index_copy_(0, where.to(codes.device), again.errors().to(codes.device))
    again.close()
    return codes


def reevaluate_wide(env, bad, fit, max_steps=EPISODE_CAP, chunk=100):
    """The overflow fallback (see ``reevaluate``): the creatures of ``bad`` again, from reset, in the wide build.  Writes their
    fitness into ``fit`` and returns the mask of creatures that overflowed even there."""
    codes = _resolve_handover(env, reevaluate(env, bad, fit, wide=True, max_steps=max_steps, chunk=chunk), fit, True, max_steps, chunk)
    env.last_error_codes = codes   # (REM2D_ERR_* of the wide run, [N]: what SolverOverflow reports for the creatures still unresolved)
    return codes != 0


def _resolve_handover(env, err, fit, wide, max_steps, chunk):
    """``err``: REM2D_ERR_* codes [N] of a run in the env's own (wide=None) or the wide build.  The creatures that carry
    REM2D_ERR_HANDOVER are evaluated again on per-step launches of that build; returns the codes with theirs replaced by the second
    run's, and logs the event on the env (``env.last_handover``: population indices).  A hand-over bit that survives per-step
    launches cannot exist (they have no hand-over); if it does, HandoverError."""
    hand = (err & _lib.ERR_HANDOVER) != 0
    if not bool(hand.any()):
        return err
    idx = torch.nonzero(hand).flatten().cpu().tolist()
    env.last_handover = sorted(set(getattr(env, "last_handover", [])) | set(idx))
    import warnings
    warnings.warn("%d creature(s) were flagged REM2D_ERR_HANDOVER by a step train launch (first: %s): evaluating them again on "
                  "per-step launches" % (len(idx), idx[:8]))
    again = reevaluate(env, hand, fit, wide=wide, options={"fuse_velpost": 1}, max_steps=max_steps, chunk=chunk)
    if bool(((again & _lib.ERR_HANDOVER) != 0).any()):
        raise HandoverError(int(((again & _lib.ERR_HANDOVER) != 0).sum()), " even on per-step launches")
    return torch.where(hand, again, err)


def run_episode_masked(env, max_steps=EPISODE_CAP, chunk=100, compact=True, fallback=True):
    """run_episode without the verdict: (fitness [N] float64, unresolved [N] bool).  ``unresolved`` marks the creatures
    whose fitness is NOT what the reference's Box2D computes: they overflowed the default build's contact capacity and
    (with ``fallback``) the wide build's as well.  Never raises on overflow -- what a sharded job calls before its
    collective (a rank that raised here would leave the others waiting in the all-gather).
    Creatures a step train flagged REM2D_ERR_HANDOVER are ALWAYS resolved here (whatever ``fallback`` says): evaluated again on
    per-step launches of the same build, never sent to the wide build for it, never penalised; ``env.last_handover`` lists them."""
    policy, env.on_handover = env.on_handover, "flag"   # (the episode goes on for everybody else; the flagged ones are dealt with below)
    try:
        _episode(env, max_steps, chunk, compact)
        fit = env.fitness.clone()
        err = env.errors().to(torch.int32)
    finally:
        env.on_handover = policy
    env.last_handover = []
    if env.handover_failures(clear=True) or bool(((err & _lib.ERR_HANDOVER) != 0).any()):
        err = _resolve_handover(env, err, fit, None, max_steps, chunk)
    bad = (err & _lib.ERR_CAPACITY) != 0
    env.last_error_codes = err
    env.last_overflow = torch.nonzero(bad).flatten().cpu().tolist()   # population indices that needed the fallback
    if fallback and bool(bad.any()):
        bad = reevaluate_wide(env, bad, fit, max_steps, chunk)
    return fit, bad


def apply_penalty(fit, bad, warn=True):
    """The defined verdict on creatures without a valid fitness (``bad``: bool mask): UNRESOLVED_FITNESS, in place, with a
    warning that names them.  Returns their indices."""
    idx = torch.nonzero(torch.as_tensor(bad)).flatten().cpu().tolist()
    if idx:
        fit[torch.as_tensor(idx, dtype=torch.long, device=fit.device)] = UNRESOLVED_FITNESS
        if warn:
            import warnings
            warnings.warn("%d creature(s) overflowed even the wide build's contact capacity (first: %s): fitness set to "
                          "UNRESOLVED_FITNESS = %s" % (len(idx), idx[:8], UNRESOLVED_FITNESS))
    return idx


def run_episode(env, max_steps=EPISODE_CAP, chunk=100, on_error="fallback", compact=True):
    """Advance a BatchedModular2D until every creature's fitness is final (or max_steps).
    Returns fitness[N] (float64 tensor on the env's device).  Engine overflows (see SolverOverflow) are not silent and,
    by default, not fatal: on_error="fallback" re-evaluates the flagged creatures in the wide build (reevaluate_wide) so
    that every individual gets the fitness Box2D would give, and raises only for creatures that overflow even that;
    on_error="penalty" (what the EA loops use) does the same but never raises: a creature beyond the wide build gets
    UNRESOLVED_FITNESS, a warning names it and ``env.last_unresolved`` lists it;
    "raise" / "warn" / "ignore" judge the default build's flags without a second attempt.  compact: in worlds created
    with REM2D_FLAG_SKIP_FROZEN (the bodies of finished creatures are nobody's business any more) the survivors are
    moved into smaller worlds between chunks once most of a world has finished (BatchedModular2D.compact) -- same
    fitness, a shorter episode."""
    if on_error == "penalty":
        fit, bad = run_episode_masked(env, max_steps, chunk, compact, fallback=True)
        env.last_unresolved = apply_penalty(fit, bad)
        return fit
    if on_error == "fallback":
        fit, bad = run_episode_masked(env, max_steps, chunk, compact, fallback=True)
        env.last_unresolved = torch.nonzero(bad).flatten().cpu().tolist()
        if bool(bad.any()):
            idx = torch.nonzero(bad).flatten().cpu().tolist()
            raise SolverOverflow(idx, env.last_error_codes[bad].cpu().tolist())   # (the creatures' real REM2D_ERR_* bits of the wide run)
        return fit
    _episode(env, max_steps, chunk, compact)
    check_errors(env, on_error)   # (REM2D_ERR_HANDOVER raises HandoverError in "raise" and "warn"; so does env.fitness below)
    return env.fitness.clone()


def evaluate_population(individuals, tree_depth=None, env=None, max_steps=EPISODE_CAP, workers=None, on_error="fallback",
                        **env_kw):
    """Batched stand-in for ``toolbox.map(toolbox.evaluate, population)``: list of floats.
    The genotype -> phenotype step runs on ``workers`` host processes (encode.encode_population).
    on_error: what to do when a creature overflowed the engine's contact capacity ("fallback": re-evaluate it in the
    wide build and raise only if even that overflows, the default | "penalty": the same without ever raising, see
    run_episode | "raise" | "warn" | "ignore")."""
    from .encode import encode_population
    from .env import BatchedModular2D
    own = env is None
    if own:
        # fitness is all that leaves this function: creatures whose fitness is final need no more steps
        env_kw.setdefault("flags", _lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN)
        env = BatchedModular2D(**env_kw)
    env.trees = env.robots = None
    from .encodings.lsystem import LSystem
    if individuals and all(type(ind.genome) is LSystem for ind in individuals) and \
            len({(ind.genome.treeDepth, ind.genome.maxModules) for ind in individuals}) == 1:
        # L-system genomes: native compiler (no fork, ~30 us per individual incl. reading the objects)
        from .encode import encode_lsystem_native
        batches = encode_lsystem_native(individuals, n_threads=workers or 0)
    elif individuals and all(_is_ff_network(ind.genome) for ind in individuals) and \
            len({ind.genome.maxModules for ind in individuals}) == 1:
        # network genomes (feed-forward CPPN): the NN queries that grow the tree run natively as well
        from .encode import encode_network_native
        depths = {tree_depth} if tree_depth is not None else {ind.tree_depth for ind in individuals}
        if len(depths) == 1:
            batches = encode_network_native(individuals, depths.pop(), n_threads=workers or 0)
        else:   # individuals with their own tree depths: the node-by-node compiler expresses each with its own
            from .encode import encode_trees_native
            batches = encode_trees_native(individuals, tree_depth, n_threads=workers or 0)
    else:
        # every other encoding: python hands out the phenotype trees, the native compiler builds the creatures
        from .encode import encode_trees_native
        batches = encode_trees_native(individuals, tree_depth, n_threads=workers or 0)
    env._upload(batches, len(individuals))
    fit = run_episode(env, max_steps, on_error=on_error).cpu().tolist()
    if own:
        env.close()
    return fit


def _is_ff_network(genome):
    from .encodings.network import FeedForwardCPPN, NNEncoding
    return type(genome) is NNEncoding and type(genome.nn_g) is FeedForwardCPPN


def shard_range(n, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank` (rank r owns [r*ceil(n/W), (r+1)*ceil(n/W)))."""
    per = math.ceil(n / world_size)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard_balanced(cost, world_size):
    """Cost-balanced shards (what replaces the dynamic balancing of REM2D_main.py:256-262's pool.map): the individuals in
    descending order of a static cost key (ties: population order -- every rank computes the same deal) are dealt snake-wise
    over the ranks, 0 .. W-1, W-1 .. 0, 0 ..  Returns int64 [W][ceil(n / W)]: row r = the population indices of rank r in the order
    it evaluates them, padded with -1.  Episodes end early for creatures that fall or get caught (Modular2DEnv.py:642-649), which
    no static key sees; what the key evens out is the per-step cost (lanes, joint rounds, solver work: all grow with the bodies)."""
    cost = np.asarray(cost)
    n, W = len(cost), int(world_size)
    per = math.ceil(n / W) if n else 0
    order = np.argsort(-cost.astype(np.int64), kind="stable")
    out = np.full((W, per), -1, dtype=np.int64)
    k = np.arange(n)
    rnd, pos = k // W, k % W
    rank = np.where(rnd % 2 == 0, pos, W - 1 - pos)
    out[rank, rnd] = order
    return out


def shard_costs(cost, index):
    """Predicted cost of every rank under the deal `index` (shard_balanced): float64 [W]."""
    cost = np.asarray(cost, dtype=np.float64)
    return np.array([cost[row[row >= 0]].sum() for row in np.asarray(index)])


def all_gather_fitness(local, n_total, group=None, flags=None, index=None):
    """One all_gather of float64 fitness scalars: [per] -> [n_total] on every rank.  `local` is padded to ceil(n/W) so
    that the collective is a single equal-sized all_gather_into_tensor.  float64 like the reference's python floats
    (REM2D_main.py:372-375 compares ``reward + (10000 - i) / 10000`` in doubles): a sharded and a single-GPU run return
    the same values, so tournament winners cannot depend on the world size.  8 MiB at 1 M individuals.
    flags: optional bool / int [per] mask of the rank's creatures without a valid fitness (run_episode_masked); it rides
    in the same collective as a second column and the call returns (fitness [n_total], flags [n_total] bool) -- every
    rank learns about every rank's failures AFTER the collective and can raise in step.
    index: the deal of shard_balanced (int64 [W][per], -1 padding) when the ranks hold dealt instead of contiguous shards:
    rank r's j-th value belongs to individual index[r][j]; the result is in population order either way."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = math.ceil(n_total / world)
    cols = 1 if flags is None else 2
    buf = torch.zeros(cols, per, dtype=torch.float64, device=local.device)
    buf[0, :local.numel()] = local.to(torch.float64)
    if flags is not None:
        buf[1, :local.numel()] = torch.as_tensor(flags, device=local.device).to(torch.float64)
    out = torch.empty(world * cols * per, dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, buf.reshape(-1), group=group)
    out = out.view(world, cols, per)
    if index is None:
        fit = out[:, 0].reshape(-1)[:n_total]
        bad = out[:, 1].reshape(-1)[:n_total] != 0 if flags is not None else None
    else:   # dealt shards: scatter back to population order
        idx = torch.as_tensor(np.asarray(index), dtype=torch.long, device=out.device).reshape(-1)
        keep = idx >= 0
        fit = torch.zeros(n_total, dtype=torch.float64, device=out.device)
        fit[idx[keep]] = out[:, 0].reshape(-1)[keep]
        bad = None
        if flags is not None:
            bad = torch.zeros(n_total, dtype=torch.bool, device=out.device)
            bad[idx[keep]] = out[:, 1].reshape(-1)[keep] != 0
    if flags is None:
        return fit
    return fit, bad


def evaluate_population_sharded(n_total, local_eval, group=None, device=None, on_error="raise", cost=None):
    """Shard [0, n_total) over the job's ranks, evaluate the local block with
    ``local_eval(lo, hi) -> tensor[hi-lo]`` (or ``-> (tensor, unresolved mask)``, run_episode_masked) and all-gather.
    cost (optional, [n_total], the same on every rank): a static cost key per individual -- the shards are then dealt by
    shard_balanced instead of cut contiguously, ``local_eval(indices)`` gets the rank's population indices (int64 array) and
    ``evaluate_population_sharded.last_shard_cost`` holds the predicted cost per rank; the result is in population order.
    Returns fitness[n_total] (float64).  The mask rides in the same collective, so every rank learns about every rank's
    unresolved creatures AFTER it: with on_error="raise" (the library default -- the reference has no contact cap and would
    have produced a fitness) every rank raises SolverOverflow in step; with "penalty" (an EA loop's opt-in) they get
    UNRESOLVED_FITNESS on every rank alike, every rank warns, ``evaluate_population_sharded.last_unresolved`` lists them and
    the job goes on."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    index = None
    if cost is not None:
        index = shard_balanced(cost, world)
        evaluate_population_sharded.last_shard_cost = shard_costs(cost, index)
        local = local_eval(index[rank][index[rank] >= 0])
    else:
        lo, hi = shard_range(n_total, rank, world)
        local = local_eval(lo, hi)
    mask = None
    if isinstance(local, tuple):
        local, mask = local
    local = torch.as_tensor(local, dtype=torch.float64, device=device if device is not None else None)
    if mask is None:
        mask = torch.zeros(local.numel(), dtype=torch.bool, device=local.device)
    fit, bad = all_gather_fitness(local, n_total, group, flags=torch.as_tensor(mask, device=local.device), index=index)
    evaluate_population_sharded.last_unresolved = []
    if bool(bad.any()):
        if on_error != "penalty":
            idx = torch.nonzero(bad).flatten().cpu().tolist()
            raise SolverOverflow(idx, [_lib.ERR_SOLVER_OVERFLOW] * len(idx))
        evaluate_population_sharded.last_unresolved = apply_penalty(fit, bad, warn=True)
    return fit
