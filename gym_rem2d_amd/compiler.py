"""Tree -> rigid-body layout -> SoA morphology batch.

Restates the reference's robot construction (``Modular2DEnv.create_robot``
``gym_rem2D/envs/Modular2DEnv.py:517-563`` and ``create_component`` ``:425-475``) without
Box2D: modules emit their bodies/joints into a :class:`CreatureBuilder`, which narrows
every value to binary32 at the point where pybox2d would (``CreateDynamicBody`` /
``revoluteJointDef`` arguments, ``body.position`` / ``body.angle`` read-backs), and the
result is packed into the ``[env][lane]`` arrays the batched stepper uploads once per
reset.  One lane = one rigid body; lane ``s >= 1`` also owns the revolute joint that ties
body ``s`` to ``parent[s]`` and the controller of the node that created it.
"""
import math

import numpy as np

from .modules import ConnectionSite, _Vec

TERRAIN_HEIGHT = 600 / 30.0 / 4      # VIEWPORT_H/SCALE/4 (Modular2DEnv.py:59)
SPAWN = (5, TERRAIN_HEIGHT + 2, 0)   # root position (Modular2DEnv.py:429-432)

SHAPE_NONE, SHAPE_BOX, SHAPE_CIRCLE = 0, 1, 2
MAX_LANES = 64

MORPH_I32 = ("shape", "parent", "jround")
MORPH_F32 = ("hx", "hy", "x", "y", "angle", "ax", "ay", "bx", "by", "torque", "lower", "upper")
MORPH_F64 = ("amp", "phase", "freq", "offset", "istate")


def f32(v):
    """Round a python double to binary32 and widen it back (the SWIG boundary)."""
    return float(np.float32(v))


class _Pos:
    """``b2Vec2``-like read-only pair: supports ``.x``/``.y`` and ``[0]``/``[1]``."""
    __slots__ = ("x", "y")

    def __init__(self, x, y):
        self.x, self.y = x, y

    def __getitem__(self, i):
        return (self.x, self.y)[i]

    def __iter__(self):
        return iter((self.x, self.y))

    def __repr__(self):
        return "(%r, %r)" % (self.x, self.y)


class BodyView:
    """Stands in for the ``b2Body`` proxies the reference stores in ``node.component`` and
    ``robot.components``.  Before upload it reports the construction pose; once bound to a
    world it reads the live pose of its lane."""

    def __init__(self, slot, shape, hx, hy, x, y, angle):
        self.slot, self.shape, self.hx, self.hy = slot, shape, hx, hy
        self._x, self._y, self._angle = x, y, angle
        self._live = None  # callable slot -> (x, y, angle) once bound
        self.color1 = self.color2 = None

    @property
    def position(self):
        if self._live is not None:
            x, y, _ = self._live(self.slot)
            return _Pos(x, y)
        return _Pos(self._x, self._y)

    @property
    def angle(self):
        if self._live is not None:
            return self._live(self.slot)[2]
        return self._angle


class JointView:
    """Stands in for the ``b2RevoluteJoint`` proxies in ``robot.joints``."""

    def __init__(self, index, a, b):
        self.index, self.bodyA, self.bodyB = index, a, b
        self.motorSpeed = 0.0

    @property
    def angle(self):
        return f32(np.float32(self.bodyB.angle) - np.float32(self.bodyA.angle))


class CreatureBuilder:
    """Records the Box2D objects one ``reset`` would create for a creature."""

    LOWER = f32(-math.pi / 2)   # revoluteJointDef lowerAngle/upperAngle (module_utility.py:27-28)
    UPPER = f32(math.pi / 2)

    def __init__(self):
        self.bodies = []   # BodyView, creation order
        self.joints = []   # dict per joint, creation order; joint k ties body k+1 to its parent
        self.cmap = lambda v: (0.5, 0.5, 0.5, 1.0)  # colour lookups of the reference are render-only

    def _add(self, shape, hx, hy, x, y, angle):
        if len(self.bodies) >= MAX_LANES:
            raise ValueError("creature exceeds %d bodies" % MAX_LANES)
        b = BodyView(len(self.bodies), shape, f32(hx), f32(hy), f32(x), f32(y), f32(angle))
        self.bodies.append(b)
        return b

    def add_box(self, hx, hy, x, y, angle):
        """world.CreateDynamicBody(position, angle, fixtures=polygonShape(box=(hx, hy)))"""
        return self._add(SHAPE_BOX, hx, hy, x, y, angle)

    def add_circle(self, r, x, y, angle):
        """world.CreateDynamicBody(position, angle, fixtures=b2CircleShape(radius=r))"""
        return self._add(SHAPE_CIRCLE, r, 0.0, x, y, angle)

    def add_revolute(self, parent, child, anchor_a, anchor_b, torque, node=None):
        """world.CreateJoint(revoluteJointDef(..., enableMotor, enableLimit, +-pi/2))"""
        ctrl = node.controller.params() if node is not None and node.controller is not None else (0.0,) * 5
        j = dict(parent=parent.slot, child=child.slot,
                 ax=f32(anchor_a[0]), ay=f32(anchor_a[1]), bx=f32(anchor_b[0]), by=f32(anchor_b[1]),
                 torque=f32(torque), lower=self.LOWER, upper=self.UPPER, ctrl=ctrl)
        self.joints.append(j)
        return JointView(len(self.joints) - 1, parent, child)


def island_joint_order(n_bodies, joints):
    """Order in which ``b2World::Solve``'s depth-first island build lists a creature's joints
    (SURVEY.md A.3): seed = last created body, joint edges head-inserted per body.
    ``joints``: list of (bodyA, bodyB) in creation order.  Returns joint indices."""
    edges = [[] for _ in range(n_bodies)]
    for k, (a, b) in enumerate(joints):
        edges[a].insert(0, k)
        edges[b].insert(0, k)
    order, jflag, bflag = [], [False] * len(joints), [False] * n_bodies
    for seed in range(n_bodies - 1, -1, -1):
        if bflag[seed]:
            continue
        stack = [seed]
        bflag[seed] = True
        while stack:
            b = stack.pop()
            for k in edges[b]:
                if jflag[k]:
                    continue
                a, c = joints[k]
                other = c if a == b else a
                order.append(k)
                jflag[k] = True
                if bflag[other]:
                    continue
                stack.append(other)
                bflag[other] = True
    return order


def joint_rounds(n_bodies, joints):
    """Earliest parallel round of every joint such that joints sharing a body keep their
    island order.  Joints of one round touch disjoint bodies, so solving a round in
    parallel is bit-identical to Box2D's sequential sweep."""
    order = island_joint_order(n_bodies, joints)
    last = [-1] * n_bodies          # last round that touched each body
    rounds = [0] * len(joints)
    for k in order:
        a, b = joints[k]
        r = max(last[a], last[b]) + 1
        rounds[k] = r
        last[a] = last[b] = r
    return rounds, order


def pipeline_schedule(n_bodies, joints, rounds):
    """Modulo schedule of the velocity iterations (one tick = a joint slot, then a contact slot).

    Joint k of iteration t fires at tick ``rounds[k] + t*P``; body b's contacts of iteration t at
    ``offC[b] + t*P`` with ``offC[b]`` at or after the last joint slot that touches b.  ``P`` is the smallest
    period for which joint k of iteration t+1 starts strictly after the contact slots (iteration t)
    of both of its bodies.  Every pair of operations that share a body then runs in Box2D's
    sequential order (joints in island order, then contacts, iteration after iteration), so the
    pipelined sweep is bit-identical to the sequential one while the chain of joints no longer
    serialises a whole iteration."""
    last = [0] * n_bodies
    first = [None] * n_bodies
    for k, (a, b) in enumerate(joints):
        for x in (a, b):
            last[x] = max(last[x], rounds[k])
            first[x] = rounds[k] if first[x] is None else min(first[x], rounds[k])
    period = 1
    for k, (a, b) in enumerate(joints):
        period = max(period, max(last[a], last[b]) + 1 - rounds[k])
    # Body b's contact slot may sit anywhere in [last[b], first[b] + period - 1] (after its last joint of
    # this iteration, before its first joint of the next).  Where that window allows it, use the phase
    # period-1 so that the contact slots of a whole wavefront coincide and the other ticks skip them.
    offC = list(last)
    for b in range(n_bodies):
        if first[b] is None:
            continue
        v = last[b] + ((period - 1 - last[b]) % period)
        if v <= first[b] + period - 1:
            offC[b] = v
    return offC, period


class CreatureSpec:
    """One creature's bodies/joints after construction."""

    def __init__(self, builder, node_slots):
        self.bodies = builder.bodies
        self.joints = builder.joints
        self.node_slots = node_slots  # node index in tree.nodes -> lane or -1
        pairs = [(j["parent"], j["child"]) for j in self.joints]
        self.rounds, self.island_order = joint_rounds(len(self.bodies), pairs)
        self.offC, self.period = pipeline_schedule(len(self.bodies), pairs, self.rounds)

    @property
    def n_bodies(self):
        return len(self.bodies)


def build_creature(nodes, module_list, terrain_height=TERRAIN_HEIGHT):
    """``create_robot`` (Modular2DEnv.py:517-563): roots first, then one sweep in list order;
    a node is built only if its parent was handled earlier in the sweep and got a body.
    Sets ``node.expressed`` / ``node.component`` like the reference (also for nodes rejected
    by the height rule, which block their sub-trees).  Returns (CreatureSpec, components,
    joints) with the latter two being ``robot.components`` / ``robot.joints``."""
    world = CreatureBuilder()
    components, joints, handled = [], [], []

    def emit(node, parent_body=None, site=None):
        module = node.module_ if node.module_ is not None else module_list[node.type]
        c, j = module.create(world, terrain_height, node=node, p_c=parent_body, module_list=module_list,
                             connection_site=site, position=list(SPAWN))
        node.expressed = True
        components.extend(c)
        joints.extend(j)

    for node in nodes:
        if node.parent == -1:
            emit(node)
            handled.append(node)
    for node in nodes:
        if node.expressed:
            continue
        parent = next((h for h in handled if h.index == node.parent and h.expressed), None)
        if parent is None or parent.component is None:
            continue
        pos, ori = parent.module_.get_global_position_of_connection_site(
            parent_component=parent.component[0], con=node.parent_connection_coordinates)
        site = ConnectionSite(_Vec(pos[0], pos[1], 0), _Vec(ori, 0, 0))
        emit(node, parent.component[0], site)
        handled.append(node)
    node_slots = [n.component[0].slot if (n.expressed and n.component is not None) else -1 for n in nodes]
    return CreatureSpec(world, node_slots), components, joints


def lanes_for(n_bodies):
    """Lanes per creature: smallest power of two >= n_bodies (>= 2)."""
    k = 2
    while k < n_bodies:
        k *= 2
    return k


class Morphology:
    """SoA batch of creature layouts in ``[env][lane]`` order (lane fastest), the upload
    format of ``rem2d_world_reset`` (include/rem2d.h).  Unused lanes have shape 0."""

    def __init__(self, n_envs, lanes):
        self.n_envs, self.lanes = int(n_envs), int(lanes)
        n = self.n_envs * self.lanes
        self.arrays = {}
        for k in MORPH_I32:
            self.arrays[k] = np.zeros(n, dtype=np.int32)
        for k in MORPH_F32:
            self.arrays[k] = np.zeros(n, dtype=np.float32)
        for k in MORPH_F64:
            self.arrays[k] = np.zeros(n, dtype=np.float64)
        self.arrays["parent"][:] = -1
        self.n_bodies = np.zeros(self.n_envs, dtype=np.int32)

    def __getitem__(self, k):
        if k == "n_envs":
            return self.n_envs
        if k == "lanes":
            return self.lanes
        return self.arrays[k]

    def set_creature(self, e, spec):
        K = self.lanes
        if spec.n_bodies > K:
            raise ValueError("creature with %d bodies does not fit %d lanes" % (spec.n_bodies, K))
        a = self.arrays
        lo = e * K
        for k in a:
            a[k][lo:lo + K] = -1 if k == "parent" else 0
        for b in spec.bodies:
            i = lo + b.slot
            # packed schedule: joint round | contact slot << 8 | period << 16 (joint round filled below)
            a["jround"][i] = (spec.offC[b.slot] << 8) | (spec.period << 16)
            a["shape"][i] = b.shape
            a["hx"][i], a["hy"][i] = b.hx, b.hy
            a["x"][i], a["y"][i], a["angle"][i] = b._x, b._y, b._angle
        for k, j in enumerate(spec.joints):
            i = lo + j["child"]
            a["parent"][i] = j["parent"]
            a["jround"][i] |= spec.rounds[k]
            for f in ("ax", "ay", "bx", "by", "torque", "lower", "upper"):
                a[f][i] = j[f]
            a["amp"][i], a["phase"][i], a["freq"][i], a["offset"][i], a["istate"][i] = j["ctrl"]
        self.n_bodies[e] = spec.n_bodies

    @classmethod
    def from_specs(cls, specs, lanes=None):
        if lanes is None:
            lanes = lanes_for(max([s.n_bodies for s in specs] + [1]))
        m = cls(len(specs), lanes)
        for e, s in enumerate(specs):
            m.set_creature(e, s)
        return m

    @classmethod
    def replicate(cls, spec, n_envs, lanes=None):
        """n_envs copies of one creature (fixed-morphology populations, BASELINE config 2)."""
        one = cls.from_specs([spec], lanes)
        m = cls(n_envs, one.lanes)
        for k, v in one.arrays.items():
            m.arrays[k][:] = np.tile(v, n_envs)
        m.n_bodies[:] = spec.n_bodies
        return m

    def take(self, idx):
        """Sub-batch of the given env indices."""
        idx = np.asarray(idx, dtype=np.int64)
        m = Morphology(len(idx), self.lanes)
        K = self.lanes
        lanes = (idx[:, None] * K + np.arange(K)[None, :]).reshape(-1)
        for k, v in self.arrays.items():
            m.arrays[k][:] = v[lanes]
        m.n_bodies[:] = self.n_bodies[idx]
        return m

    @classmethod
    def concat(cls, parts):
        """The creatures of several batches of one lane count, one after the other."""
        parts = list(parts)
        lanes = parts[0].lanes
        if any(p.lanes != lanes for p in parts):
            raise ValueError("Morphology.concat needs batches of one lane count")
        m = cls(sum(p.n_envs for p in parts), lanes)
        for k in m.arrays:
            m.arrays[k][:] = np.concatenate([p.arrays[k] for p in parts])
        m.n_bodies[:] = np.concatenate([p.n_bodies for p in parts])
        return m

    def as_dict(self):
        d = dict(self.arrays)
        d["n_envs"], d["lanes"] = self.n_envs, self.lanes
        return d
