"""gym-shaped environments on top of the batched MI355X stepper.

``Modular2D`` keeps the reference env's surface (``gym_rem2D/envs/Modular2DEnv.py:127-173,
565-653``): ``seed(int) -> [seed]``, ``reset(tree=, module_list=) -> None``,
``step(action) -> (0, reward, done, 0)``, attributes ``action_space``, ``observation_space``,
``hardcore``, ``robot.components/joints``, ``tree_morphology``, ``wod``.  It is a batch of one.

``BatchedModular2D`` is the same environment for N creatures at once: ``reset(trees,
module_lists)`` compiles every tree into the SoA layout, groups creatures by lane count
(homogeneous waves) and uploads them; ``step(n)`` advances all of them n steps on the GPU and
returns ``reward[N]`` / ``done[N]`` torch tensors.  Rendering (pyglet, ``:655-768``) is out of
scope.
"""
import copy
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, gymshim
from .compiler import Morphology, build_creature, lanes_for
from .terrain import make_terrain
from .world import BatchedWorld

FPS = 50
WOD_SPEED = 0.04
VIEWPORT_W, VIEWPORT_H, SCALE = 800, 600, 30.0


class ModularRobotBox2D:
    """robot.components / robot.joints containers (Modular2DEnv.py:69-82)."""

    def __init__(self):
        self.components, self.joints = [], []

    def add_components(self, components, joints=None):
        self.components.extend(components)
        if joints is not None:
            self.joints.extend(joints)
        return self


class WallOfDeath:
    def __init__(self, speed):
        self.position = 0.0
        self.speed = speed

    def update(self):
        self.position += self.speed


def _uniform(m):
    """Every creature of the batch has the same tree and solver schedule (fixed-morphology population)."""
    n, K = m.n_envs, m.lanes
    return all(bool((m.arrays[k].reshape(n, K) == m.arrays[k][:K]).all()) for k in ("shape", "parent", "jround"))


_GROUP_STREAMS = {}   # device -> the step-group streams of this process


def group_streams(device, n):
    """The n - 1 HIP streams step groups 1 .. n-1 run on (group 0 runs on the caller's stream), shared by every
    BatchedModular2D of the process: HIP maps streams onto a few hardware queues and streams that share a queue serialise,
    so a second env must not bring streams of its own (65 536 CPPN creatures: 24.8 M env-steps/s on the 6th-8th stream
    of a process, 45 M on the first three)."""
    dev = torch.device(device if device is not None else "cuda")
    index = dev.index if dev.index is not None else torch.cuda.current_device()   # (None, "cuda", "cuda:0": ONE pool)
    pool = _GROUP_STREAMS.setdefault(index, [])
    while len(pool) < n - 1:
        pool.append(torch.cuda.Stream(device=device))
    return [None] + pool[:max(0, n - 1)]


class BatchedModular2D:
    MAX_WORLD_LANES = 1 << 22   # rem2d_world_create refuses ~5 M lanes and more (32-bit lane offsets)
    BIG_POPULATION = 131072     # creatures per GPU from which the 128-lane tiles of the velocity kernel pay (round 4: 2 joint
                                # register sets at 4 wavefronts per SIMD, 18.6 active lanes; profiles/r04_sweep_population_shape.txt)
    REBALANCE_EVERY = 50        # env-steps between two re-orderings of a mixed population by current cost (see __init__)
    TRAIN128_MAX = 131072       # creatures per GPU up to which the 128-lane step train is used instead of per-step launches (profiles/
                                # r06_train128.txt; on the round's final build: 131 072: 71.4 vs 71.3 M env-steps/s, 163 840: 76.1 vs 78.6 M,
                                # 196 608: 75.5 vs 82.6 M -- the 4-phase loop and -Os helped the per-step launches more)
    TRAIN128_UNIFORM = False    # uniform populations on the static 128-lane shape keep per-step launches on three step groups: 65 536 8-module
                                # chains 180.7 M against 177.0 M as a train (and 142.5 M as a 64-lane train) -- profiles/r06_train128.txt

    def __init__(self, hardcore=False, flat=False, seed=4, device=None, flags=None, wide=False, options=None, on_handover="raise"):
        # pybox2d's b2World() defaults: continuousPhysics on, sleeping on
        # options: launch options for every world ({name: value}, _lib.OPTIONS -- launch forms, never results), e.g.
        # {"fuse_velpost": 1} = per-step launches instead of the step train.  on_handover: what step() / fitness / errors() do when a
        # step train reported failed hand-overs (rem2d_world_handover_failures): "raise" _lib.HandoverError (nothing computed from
        # such a state is handed out silently), or "flag" = leave it to the caller, who reads errors() & ERR_HANDOVER
        # (evaluate.run_episode: re-evaluates those creatures on per-step launches)
        # wide: the worlds live in librem2d_wide.so (32 pair slots / 12 solver slots per body; evaluate.run_episode re-runs
        # there the creatures that overflowed the default build's slots)
        from . import _lib
        self.wide = wide if wide == "fma" else bool(wide)   # ("fma": the labelled -ffp-contract=fast tolerance mode, not bit-exact)
        self.hardcore, self.flat = hardcore, flat
        self.flags = _lib.FLAG_CONTINUOUS if flags is None else flags
        self.device = device
        self._seed = seed
        self.terrain = None
        self.worlds = []      # list of (BatchedWorld, env index tensor)
        self.n_envs = 0
        self.trees = None
        self.robots = None
        self._reward = self._done = None
        # REM2D_MERGED_LAUNCH=0: step every lane bucket on its own stream instead of one merged grid
        self.merged_launch = os.environ.get("REM2D_MERGED_LAUNCH", "1") != "0"
        self.step_groups = int(os.environ.get("REM2D_STEP_GROUPS", "0"))  # 0 = automatic
        # launch shape of the velocity kernel (rem2d_world_set_tile_shape): None = automatic, unless the experiment
        # override REM2D_TILE_SHAPE fixes it (_lib.env_tile_shape).  Up to ~150 000 creatures a step is bound by its chain of stragglers and the
        # 64-lane tiles (4 wavefronts per SIMD) win; beyond that the chip's instruction issue saturates and the 256-lane
        # tiles (2.4x fewer wave-instructions) do: 62.1 vs 58.0 M env-steps/s at 196 608 creatures.
        self.tile_shape = None
        self.groups, self.group_streams = [], []
        self._group_args = None
        self.use_graph = os.environ.get("REM2D_GRAPH", "0") == "1"   # replay every step call as a hipGraph
        # Creature order by current cost: every `rebalance_every` env-steps the creatures that used every position iteration in
        # the last step are moved to the front of their world's order (a stable partition: the static schedule order survives
        # within both classes), so that they share velocity tiles and position blocks -- a tile costs what its most expensive
        # creature costs.  Made on the device by the library itself (launch option `rebalance`, rem2d_rebalance_kernel: one
        # small launch per world every N steps).  -1: automatic (50 for mixed populations: +3.3 % on config 3, +3 % on the
        # 131 072-creature generation against REM2D_FLAG_RETILE, which it replaces as the policy; +0.5 % on config 4; -0.8 % at
        # 1 M creatures), 0: off.  REM2D_REBALANCE overrides (experiments); rebalance() does the same from the host.
        self.rebalance_every = int(os.environ.get("REM2D_REBALANCE", "-1"))
        self.options = dict(options or {})
        if on_handover not in ("raise", "flag"):
            raise ValueError("on_handover must be 'raise' or 'flag'")
        self.on_handover = on_handover

    def seed(self, seed=None):
        self._seed = seed
        self.terrain = None
        return [seed]

    def _terrain(self):
        if self.terrain is None:
            self.terrain = make_terrain(self._seed, hardcore=self.hardcore, flat=self.flat)
        return self.terrain

    # ---- reset from phenotype trees (reference-shaped) ----
    def reset(self, trees, module_lists=None):
        """trees: list of Tree; module_lists: list (or one shared list) of module prototypes.
        Like the reference, each tree is deep-copied so that the env owns controller state."""
        if module_lists is None or (len(module_lists) > 0 and not isinstance(module_lists[0], (list, tuple))):
            module_lists = [module_lists] * len(trees)
        self.trees, self.robots, specs = [], [], []
        for tree, ml in zip(trees, module_lists):
            t = copy.deepcopy(tree)
            spec, comps, joints = build_creature(t.getNodes(), ml)
            self.trees.append(t)
            self.robots.append(ModularRobotBox2D().add_components(comps, joints))
            specs.append(spec)
        self.reset_specs(specs)
        self._bind_views()

    def reset_specs(self, specs):
        groups = {}
        for e, s in enumerate(specs):
            groups.setdefault(lanes_for(s.n_bodies), []).append(e)
        batches = []
        for lanes in sorted(groups):
            # creatures of one wave run in lockstep: keep waves homogeneous in joint rounds / size
            # (the most complex first: their wavefronts are the long ones and should be dispatched first)
            idx = sorted(groups[lanes], key=lambda e: (specs[e].period, max(specs[e].rounds, default=-1), specs[e].n_bodies),
                         reverse=os.environ.get("REM2D_SORT_DESC", "1") != "0")
            batches.append((Morphology.from_specs([specs[e] for e in idx], lanes), idx))
        self._upload(batches, len(specs))

    def reset_morphology(self, morph):
        """Fast path: a precompiled SoA batch (one lane count)."""
        self.trees = self.robots = None
        self._upload([(morph, list(range(morph.n_envs)))], morph.n_envs)

    def _upload(self, batches, n_envs):
        for w, _ in self.worlds:
            w.close()
        self.worlds = []
        self._world_morph = []   # host-side layout of every world's creatures (compact() re-plans tiles from it)
        self._uploaded = [(m, np.asarray(idx, dtype=np.int64)) for m, idx in batches]   # (evaluate.run_episode's fallback)
        self._compacted = False
        self._inactive = set()   # worlds compact() found without a single open fitness
        self._tile_shape_used = None
        self._group_args = None
        self.n_envs = n_envs
        self.streams = []
        # Step groups: a step is a chain of four launches, each as long as its slowest wavefront; independent parts of the
        # population on their own streams let one part's tail run under another part's kernels.  (Creatures are
        # independent, so any split is legal.)  What counts is the number of 64-lane blocks and how long a step is: below
        # ~3 000 blocks the chip is not full anyway (4 096 4-module chains: 23.0 M env-steps/s with one group, 20.3 M with
        # two); mixed or wide-creature populations, whose steps take more than a millisecond, gain up to four groups
        # (65 536 L-system creatures, 7 790 blocks: 26 / 36 / 39 / 40 M with 1 / 2 / 3 / 4; CPPN creatures on the
        # hardcore terrain 38.7 / 43.2 / 44.3 M with 2 / 3 / 4); small uniform creatures, whose steps are short, three
        # (65 536 8-module chains, 8 192 blocks: 121 / 160 / 171 / 135 M with 1 / 2 / 3 / 4).  Never more than four streams
        # in all, the caller's included (see group_streams below).
        groups = self.step_groups
        blocks = sum(m.n_envs * m.lanes for m, _ in batches) / 64.0
        if groups <= 0:
            long_steps = len(batches) > 1 or max(m.lanes for m, _ in batches) >= 16
            if long_steps:
                groups = 4 if blocks >= 512 else 1   # (8 192 / 16 384 / 24 576 L-system creatures: +13 / +11 / +15 % over one)
            else:
                groups = 3 if blocks >= 6144 else (2 if blocks >= 3072 else 1)
        # Tile shape of the velocity kernel: 64-lane tiles up to ~130 000 creatures, 128-lane tiles beyond (see __init__).
        # Fixed-morphology populations (every creature the same tree: the north-star's "8-module creatures") are the
        # exception: all creatures of a tile need the same slots per iteration, so a bigger tile costs no more per
        # iteration and halves the wavefronts -- 128-lane tiles: 170 M instead of 136 M env-steps/s for 65 536 8-module
        # chains -- once the 64-lane tiles of a step group would no longer fit the chip at once.
        shape = self.tile_shape
        if shape is None and _lib.env_tile_shape() is None:
            shape = 1 if n_envs >= self.BIG_POPULATION else 3
            if shape == 3 and blocks / groups > 2048 and all(_uniform(m) for m, _ in batches):
                shape = 4   # (128-lane tiles with the static phase -> set map: nothing to rotate in a uniform population)
        self._tile_shape_used = shape
        # (experiment override, host layer only: a tile shape per lane bucket, "lanes:shape,..." -- e.g. the light buckets on 128-lane
        # tiles beside the 16-lane bucket on 64-lane ones in ONE launch: the launch takes the kernel of the largest shape)
        by_lanes = {}
        for item in os.environ.get("REM2D_TILE_SHAPE_BY_LANES", "").split(","):
            if ":" in item:
                by_lanes[int(item.split(":")[0])] = int(item.split(":")[1])
        self._tile_shape_by_lanes = by_lanes
        # REM2D_FLAG_RETILE (the position kernel deals the creatures anew in every step, in arrival order) was round 3's policy
        # for >= 98 304 creatures; the stable re-ordering every 50 steps does better there and also pays at 65 536
        # (profiles/r04_lane_fill_experiments.txt), so the flag is an experiment override now (REM2D_RETILE=1)
        retile = os.environ.get("REM2D_RETILE") == "1"
        # The step train (the library's default launch form for 64-lane tiles, REM2D_OPT_FUSE_VELPOST = 2: all steps of a call in
        # one launch, block-steps handed from workgroup to workgroup) is ONE in-order train: it wants the whole population in one
        # group (config 3: 64.8 M env-steps/s with one group, 59 M with two, 39 M with four -- profiles/r05_step_train.txt).
        # Round 6: the 128-lane tile shapes have a train of their own (rem2d_step_train128_kernel: an item = a tile's two blocks).  It
        # wins while a step is bound by the chain of its launches and loses once the chip's instruction issue saturates: shape 1
        # up to TRAIN128_MAX creatures (beyond: per-step launches on four step groups, as before); shape 4 (uniform populations) --
        # see TRAIN128_UNIFORM.
        opts = dict(_lib.env_options(), **self.options)
        eff_shape = shape if shape is not None else _lib.env_tile_shape()
        self._launch_options = {}
        if "fuse_velpost" not in opts and ((eff_shape == 1 and n_envs > self.TRAIN128_MAX) or (eff_shape == 4 and not self.TRAIN128_UNIFORM)):
            self._launch_options["fuse_velpost"] = 1   # (per-step launches; no result depends on it)
            opts["fuse_velpost"] = 1
        train = eff_shape in (3, 1, 4) and not retile and \
            opts.get("fuse_velpost", 2) == 2 and opts.get("pipeline", 3) == 3 and opts.get("debug", 0) == 0
        if train and self.step_groups <= 0:
            groups = 1      # ... unless its lane buckets, cut into worlds of <= MAX_WORLD_LANES lanes, are more than one launch takes

            def worlds_per_group(g):
                return sum(-(-(-(-m.n_envs // g)) // max(1, self.MAX_WORLD_LANES // m.lanes)) for m, _ in batches)
            while groups < _lib.MAX_STEP_GROUPS and worlds_per_group(groups) > _lib.MAX_WORLDS_PER_STEP:
                groups += 1
        self._world_flags = (self.flags | _lib.FLAG_RETILE) if retile else (self.flags & ~_lib.FLAG_RETILE)
        every = self.rebalance_every
        if every < 0:
            every = self.REBALANCE_EVERY if (n_envs >= 4096 and not all(_uniform(m) for m, _ in batches)) else 0
        self._rebalance_steps = 0 if retile else every
        self.groups = [[] for _ in range(groups)]
        for morph, idx in batches:
            idx = np.asarray(idx, dtype=np.int64)
            # which creatures go to which group: wavefront-sized runs of the (schedule-sorted) batch are dealt round-robin,
            # so that every group gets the same mix of simple and complex creatures and the groups reach the join at the
            # end of a step call together (contiguous parts, REM2D_GROUP_SPLIT=cut: 40.2 instead of 40.6 M on config 3)
            if morph.n_envs < 4 * groups:
                members = [np.arange(morph.n_envs)]
            elif os.environ.get("REM2D_GROUP_SPLIT") != "cut":
                run = max(1, 64 // morph.lanes)
                which = (np.arange(morph.n_envs) // run) % groups
                members = [np.nonzero(which == g)[0] for g in range(groups)]
            else:
                cuts = [morph.n_envs * g // groups for g in range(groups + 1)]
                members = [np.arange(cuts[g], cuts[g + 1]) for g in range(groups)]
            pieces = []
            for g, mem in enumerate(members):   # one world addresses its lanes with 32-bit offsets: <= MAX_WORLD_LANES
                per = max(1, self.MAX_WORLD_LANES // morph.lanes)
                for lo in range(0, len(mem), per):
                    pieces.append((g, mem[lo:lo + per]))
            for g, mem in pieces:
                part = morph if len(mem) == morph.n_envs else morph.take(mem)
                w = BatchedWorld(part.n_envs, part.lanes, self._world_flags, self.device, wide=self.wide, options=self._world_options())
                w.set_terrain(self._terrain())
                w.reset(part, tile_shape=self._tile_shape_by_lanes.get(part.lanes, shape))
                self.groups[g].append(len(self.worlds))
                self.worlds.append((w, torch.as_tensor(idx[mem], dtype=torch.long, device=w.device)))
                self._world_morph.append(part)
        self.groups = [g for g in self.groups if g]
        dev = self.worlds[0][0].device
        # the first group runs on the caller's stream: four streams in all is what the device overlaps well (a fifth costs
        # 5-25 %: 4 groups on 4 new streams 36.7 M, on the caller's + 3 new ones 40.2 M env-steps/s for config 3)
        self.group_streams = group_streams(dev, len(self.groups))
        self._reward = torch.zeros(n_envs, dtype=torch.float32, device=dev)
        self._done = torch.zeros(n_envs, dtype=torch.bool, device=dev)
        self._fitness = torch.zeros(n_envs, dtype=torch.float64, device=dev)
        self._frozen_pop = torch.zeros(n_envs, dtype=torch.int32, device=dev)
        self._steps_pop = torch.zeros(n_envs, dtype=torch.int32, device=dev)
        self._err_pop = torch.zeros(n_envs, dtype=torch.int32, device=dev)
        if len(self.worlds) > 1:
            # the kernels write reward / done straight into these population-order arrays (rem2d_world_set_outputs):
            # step() returns them without a gather per world
            for w, idx in self.worlds:
                w.set_outputs(self._reward, self._done, idx.to(torch.int32))

    def _world_options(self):
        opts = dict(getattr(self, "_launch_options", {}), **self.options)
        if self._rebalance_steps > 0:
            opts.setdefault("rebalance", self._rebalance_steps)
        return opts or None

    def handover_failures(self, clear=False):
        """Failed hand-overs of the step train over all worlds (host-side counters, no synchronisation: what the launches that have
        finished so far reported)."""
        return sum(w.handover_failures(clear) for wi, (w, _) in enumerate(self.worlds) if getattr(w, "h", None))

    def check_handover(self, sync=False):
        """Raise _lib.HandoverError if a step train of this env reported failed hand-overs (unless on_handover == "flag").  sync:
        wait for the queued launches first -- the counters only know what has finished."""
        if self.on_handover != "raise" or not self.worlds:
            return
        if sync:
            torch.cuda.synchronize(self.worlds[0][0].device)
        n = self.handover_failures()
        if n:
            raise _lib.HandoverError(n)

    def _bind_views(self):
        """Let node.component / robot.components read live poses (host read-back; API parity only)."""
        if self.robots is None:
            return
        where = {}
        for w, idx in self.worlds:
            for local, e in enumerate(idx.tolist()):
                where[e] = (w, local)
        for e, robot in enumerate(self.robots):
            w, local = where[e]

            def live(slot, w=w, local=local):
                return (float(w.view("px")[local, slot]), float(w.view("py")[local, slot]),
                        float(w.view("ang")[local, slot]))
            for b in robot.components:
                b._live = live

    # ---- step ----
    def step(self, n_steps=1):
        if not self.groups and len(self.worlds) != 1:   # compact() has retired every world: nothing left to step
            return self._reward, self._done
        self.check_handover()   # (nothing more is queued behind a launch that reported a failed hand-over)
        if len(self.worlds) == 1:
            self.worlds[0][0].step(n_steps)
        elif (self.merged_launch and len(self.groups) <= _lib.MAX_STEP_GROUPS
              and all(len(g) <= _lib.MAX_WORLDS_PER_STEP for g in self.groups)):
            # all lane buckets of a group in one grid per kernel, all groups in ONE ABI call (rem2d_groups_step): fork from
            # the caller's stream, the steps of the groups queued round-robin, join -- or the whole call replayed as a
            # hipGraph (REM2D_GRAPH=1)
            if self._group_args is None:
                arrs = [(C.c_void_p * len(g))(*[self.worlds[i][0].h for i in g]) for g in self.groups]
                sg = (_lib.StepGroup * len(self.groups))()
                for k, (g, st) in enumerate(zip(self.groups, self.group_streams)):
                    sg[k].worlds = C.cast(arrs[k], C.POINTER(C.c_void_p))
                    sg[k].n_worlds = len(g)
                    sg[k].stream = None if st is None else st.cuda_stream
                self._group_args = (sg, arrs)
            _lib.check(_lib.lib(self.wide).rem2d_groups_step(self._group_args[0], len(self.groups), int(n_steps),
                                                             self.worlds[self.groups[0][0]][0]._stream(),
                                                             _lib.STEP_GRAPH if self.use_graph else 0), self.wide)
        else:
            # fallback path (REM2D_MERGED_LAUNCH=0, or more lane buckets than one launch takes): one HIP stream per world,
            # created on first use
            cur = torch.cuda.current_stream(self.worlds[0][0].device)
            while len(self.streams) < len(self.worlds):
                self.streams.append(torch.cuda.Stream(device=self.worlds[0][0].device))
            for wi, ((w, _), st) in enumerate(zip(self.worlds, self.streams)):
                if wi in self._inactive:
                    continue
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    w.step(n_steps)
            for st in self.streams:
                cur.wait_stream(st)
        # (no synchronisation: this sees the launches that have finished -- a failure of the call just queued surfaces at the next
        # step(), or at fitness / errors(), which wait first)
        self.check_handover()
        if len(self.worlds) == 1 and not self._compacted:
            w = self.worlds[0][0]
            return w.view("reward"), w.view("done") != 0
        return self._reward, self._done   # written by the step's own kernels (set_outputs in _upload)

    def rebalance(self, pos_iters=60):
        """The host-side form of the `rebalance` launch option (which does the same on the device every N steps): give every
        world a creature order by CURRENT cost through rem2d_world_set_order -- the creatures that used all `pos_iters`
        position iterations in the last step (a joint at its limit pressed against the ground; the same creatures for many
        steps) go to the front, in their static order, the others follow in theirs.  A few small torch kernels per world,
        queued on the caller's stream like the steps; no effect on any result."""
        for wi, (w, _) in enumerate(self.worlds):
            if wi in self._inactive or (w.flags & _lib.FLAG_RETILE) or w.n_envs < 2 * (64 // min(64, w.lanes)):
                continue
            slow = w.view("positers") >= pos_iters
            w.set_order(torch.sort((~slow).to(torch.uint8), stable=True).indices, check=False)   # (a sort's indices)

    def launch_info(self):
        """(tile shape, launch form) of the first step group -- the library's own answer (rem2d_worlds_launch_info: 2 = the step
        train, 1 = velocity tiles and position iterations in one launch per step, 0 = two launches); for tools that name
        kernels, results never depend on it."""
        idx = self.groups[0] if self.groups else list(range(len(self.worlds)))
        arr = (C.c_void_p * len(idx))(*[self.worlds[i][0].h for i in idx])
        shape, fused = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib(self.wide).rem2d_worlds_launch_info(arr, len(idx), C.byref(shape), C.byref(fused)), self.wide)
        return shape.value, int(fused.value)

    def _gather(self, name, out):
        if len(self.worlds) == 1 and not self._compacted:
            return self.worlds[0][0].view(name).clone()   # a snapshot, like the multi-world path
        for wi, (w, idx) in enumerate(self.worlds):   # (creatures compact() has dropped keep the values it stored in `out`)
            if wi not in self._inactive:
                out.index_copy_(0, idx, w.view(name).to(out.dtype))
        return out.clone()   # (`out` is the persistent population-order buffer compact() writes to: callers get a snapshot)

    # ---- evaluate(): drop the creatures whose fitness is final ----
    def compact(self, min_envs=2048, max_alive=0.5):
        """Between two steps of an evaluate() episode: when at most ``max_alive`` of the creatures of a lane bucket still
        have an open fitness (REM2D_F_FROZEN == 0), the bucket's worlds (one per step group) are replaced by ONE smaller
        world that holds only those -- state moved field by field, rem2d_world_adopt -- so that the wavefronts of the
        finished creatures stop costing anything (REM2D_FLAG_SKIP_FROZEN only stops wavefronts whose creatures have ALL
        finished) and the few survivors of a long episode are stepped by one launch sequence instead of one per step
        group.  Fitness / steps / error bits of the dropped creatures stay readable through the population-order
        properties.  Buckets with fewer than ``min_envs`` creatures are left alone.  Returns the number of creatures
        still being stepped."""
        by_lanes = {}
        for wi, (w, idx) in enumerate(self.worlds):
            if wi not in self._inactive:
                by_lanes.setdefault(w.lanes, []).append(wi)
        alive_total = 0
        changed = False
        for lanes, wis in sorted(by_lanes.items()):
            keeps = [torch.nonzero(self.worlds[wi][0].view("frozen") == 0, as_tuple=False).flatten() for wi in wis]
            n_keep = sum(int(k.numel()) for k in keeps)
            n_now = sum(self.worlds[wi][0].n_envs for wi in wis)
            alive_total += n_keep
            if n_now < min_envs or n_keep > max_alive * n_now:
                continue
            changed = True
            for wi in wis:   # what the population-order properties report for the creatures that leave
                w, idx = self.worlds[wi]
                self._fitness.index_copy_(0, idx, w.view("fitness"))
                self._frozen_pop.index_copy_(0, idx, w.view("frozen"))
                self._steps_pop.index_copy_(0, idx, w.view("steps"))
                self._err_pop.index_copy_(0, idx, w.view("err"))
                self._reward.index_copy_(0, idx, w.view("reward"))
                self._done.index_copy_(0, idx, w.view("done") != 0)
            if n_keep == 0:   # nobody left: the worlds stay as they are and are not launched any more
                self._inactive.update(wis)
                continue
            part = Morphology.concat([self._world_morph[wi].take(k.cpu().numpy()) for wi, k in zip(wis, keeps) if k.numel()])
            nw = BatchedWorld(n_keep, lanes, self._world_flags, self.device, wide=self.wide, options=self._world_options())
            nw.set_terrain(self._terrain())
            for name in _lib.FIELDS:
                dst = nw.view(name)
                dim = 1 if dst.dim() == 3 else 0
                dst.copy_(torch.cat([self.worlds[wi][0].view(name).index_select(dim, k) for wi, k in zip(wis, keeps)
                                     if k.numel()], dim=dim))
            nw.adopt(part, tile_shape=self._tile_shape_used)
            new_idx = torch.cat([self.worlds[wi][1][k] for wi, k in zip(wis, keeps) if k.numel()])
            nw.set_outputs(self._reward, self._done, new_idx.to(torch.int32))
            torch.cuda.synchronize(nw.device)   # the old arenas must outlive the copies
            for wi in wis[1:]:   # (the entries stay so that world indices do not move; the arenas go)
                self._inactive.add(wi)
                self.worlds[wi][0].release()
            self.worlds[wis[0]][0].release()
            self.worlds[wis[0]] = (nw, new_idx)
            self._world_morph[wis[0]] = part
        if changed:
            self._compacted = True
            groups = [[i for i in g if i not in self._inactive] for g in self.groups]
            groups = [g for g in groups if g]
            active = [i for g in groups for i in g]
            if sum(self.worlds[i][0].n_envs for i in active) < 16384 and len(active) <= _lib.MAX_WORLDS_PER_STEP:
                groups = [active] if active else []   # too few creatures for step groups to pay: one launch sequence
            self.groups = groups
            self.group_streams = self.group_streams[:max(1, len(groups))]   # (the first is the caller's stream: None)
            self._group_args = None
        return alive_total

    @property
    def fitness(self):
        """evaluate()'s running fitness (REM2D_main.py:362-377), float64 [N]."""
        self.check_handover(sync=True)
        return self._gather("fitness", self._fitness)

    @property
    def frozen(self):
        return self._gather("frozen", self._frozen_pop)

    @property
    def steps(self):
        """env steps taken since reset, int32 [N]."""
        return self._gather("steps", self._steps_pop)

    def errors(self):
        """REM2D_ERR_* bits per creature, int32 [N] (the one read that never raises HandoverError: it is how a caller finds the
        creatures concerned)."""
        return self._gather("err", self._err_pop)

    def close(self):
        for w, _ in self.worlds:
            w.close()
        self.worlds = []


class Modular2D(gymshim.Env):
    """Single-creature facade with the reference's call signatures."""
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': FPS}
    hardcore = False

    def __init__(self, random_seed=None, device=None):
        self._device = device
        self.seed(random_seed)
        self.viewer = None
        self.tree_morphology = None
        self.robot = None
        self.world = None
        self.wod = None
        self.game_over = False
        high = np.array([np.inf] * 24)
        self.action_space = gymshim.Box(np.array([-1, -1, -1, -1]), np.array([1, 1, 1, 1]), dtype=np.float32)
        self.observation_space = gymshim.Box(-high, high, dtype=np.float32)
        self._batch = None

    def seed(self, seed=None):
        self.np_random, seed = gymshim.np_random(seed)
        self._seed_value = seed
        return [seed]

    def reset(self, tree=None, module_list=None):
        self.wod = WallOfDeath(WOD_SPEED)
        self.game_over = False
        if self._batch is not None:
            self._batch.close()
        self._batch = None
        self.tree_morphology = None
        self.robot = ModularRobotBox2D()
        if tree is None:
            return
        self._batch = BatchedModular2D(hardcore=self.hardcore, seed=self._seed_value, device=self._device)
        self._batch.reset([tree], [module_list])
        self.tree_morphology = self._batch.trees[0]
        self.robot = self._batch.robots[0]
        self.world = self._batch.worlds[0][0]
        # the reference counts the expressed controllers in every step (Modular2DEnv.py:617-630); the tree cannot change
        # between resets, so count once
        self._n_ctrl = sum(1 for n in self.tree_morphology.nodes
                           if n.controller is not None and n.expressed and n.component is not None)
        # reward / done of every step land in pinned, device-mapped host memory, written by the step's own kernels
        # (rem2d_world_set_outputs): step() needs no copy kernel and no device -> host memcpy, only the stream's completion
        self._pin_reward = torch.zeros(1, dtype=torch.float32).pin_memory()
        self._pin_done = torch.zeros(1, dtype=torch.bool).pin_memory()
        self._pin_index = torch.zeros(1, dtype=torch.int32, device=self.world.device)
        self.world.set_outputs(self._pin_reward, self._pin_done, self._pin_index)
        return

    def step(self, action):
        if self.wod:
            self.wod.update()
        if self.tree_morphology is None:
            raise Exception("no tree_morphology")
        assert self._n_ctrl - 1 == len(self.robot.joints)
        self.world.step(1)                      # one creature: straight to the C ABI, no bucket / group bookkeeping
        torch.cuda.current_stream(self.world.device).synchronize()
        r, d = float(self._pin_reward[0]), bool(self._pin_done[0])   # (host reads of the mapped buffer the kernels wrote)
        return 0, (r if not d else -100), (True if d else 0), 0

    def render(self, mode='human'):
        """The reference paints into a pyglet window (Modular2DEnv.py:655-738; pyglet is not a dependency here).
        ``mode='rgb_array'`` returns the same picture -- terrain, module boxes / circles, joint anchors, wall of death,
        camera following the root -- as an ``[H, W, 3] uint8`` array drawn with matplotlib (Agg) from a state dump of
        the current step; ``mode='human'`` is refused (no window system on this path)."""
        if mode != 'rgb_array':
            raise NotImplementedError("only mode='rgb_array' is available (no pyglet window on the accelerated path); "
                                      "statedump.record_episode + tools/render_dump.py write whole runs")
        if self._batch is None:
            raise Exception("no tree_morphology")
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        from matplotlib.patches import Circle, Polygon
        from . import statedump
        head = statedump.header(self._batch, [0])
        fr = statedump.frame(self._batch, [0], 0)
        prims = statedump.frame_to_draw_list(head, fr)[0]
        cx = fr["creatures"][0]["pose"][0][0]
        fig, ax = plt.subplots(figsize=(VIEWPORT_W / 100.0, VIEWPORT_H / 100.0), dpi=100)
        ax.plot(head["terrain"]["x"], head["terrain"]["y"], color="#356635", lw=1.5)
        for box in head["terrain"]["boxes"]:
            ax.add_patch(Polygon(box, closed=True, color="#444444"))
        for p in prims:
            if p[0] == "polygon":
                ax.add_patch(Polygon(p[1], closed=True, facecolor="#7fa6d9", edgecolor="#1f3f66"))
            elif p[0] == "circle":
                ax.add_patch(Circle(p[1], p[2], facecolor="#d9a67f", edgecolor="#66401f"))
            else:
                ax.plot([p[1][0]], [p[1][1]], "k.", ms=3)
        ax.axvline(fr["creatures"][0]["wall_of_death"], color="red", lw=1)
        half = VIEWPORT_W / SCALE / 2
        ax.set_xlim(cx - half, cx + half)
        ax.set_ylim(0, VIEWPORT_H / SCALE)
        ax.set_aspect("equal")
        ax.axis("off")
        fig.subplots_adjust(0, 0, 1, 1)
        fig.canvas.draw()
        img = np.asarray(fig.canvas.buffer_rgba())[..., :3].copy()
        plt.close(fig)
        return img

    def close(self):
        if self._batch is not None:
            self._batch.close()
            self._batch = None
