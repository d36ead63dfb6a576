#!/usr/bin/env python3
"""Capture golden vectors from the importable Python side of the reference.

Runs ONLY in the build container: imports /root/reference/ModularER_2D **in place**
(nothing is copied), with two stand-in modules for the third-party packages that are not
installed there -- `gym` (base classes + gym-0.18 seeding) and `Box2D` (a *recording* world
that stores what the reference asks pybox2d to create and hands positions/angles back
rounded to binary32, as the SWIG layer does).  The physics step itself cannot be captured
(no Box2D anywhere): these fixtures pin the boundary -- tree generation, body/joint layout,
anchors, controller sequence, terrain profile -- not `world.Step`.

Writes small numeric JSON files to tests/golden/.  The GPU box never sees the reference.
"""
import importlib.util
import json
import math
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference/ModularER_2D"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def f32(v):
    return float(np.float32(v))


# ---------------------------------------------------------------- stand-in: gym
def install_gym_stub():
    import hashlib
    import struct

    gym = types.ModuleType("gym")

    class Env:
        pass

    class Box:
        def __init__(self, low, high, dtype=None, shape=None):
            self.low, self.high, self.dtype = np.asarray(low), np.asarray(high), dtype

        def sample(self):
            return np.zeros_like(self.low, dtype=np.float32)

    def np_random(seed=None):
        if seed is None:
            seed = 0
        h = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
        h += b"\0" * (4 - len(h) % 4)
        big = sum(2 ** (32 * i) * v for i, v in enumerate(struct.unpack("%dI" % (len(h) // 4), h)))
        limbs = []
        while big > 0:
            big, mod = divmod(big, 2 ** 32)
            limbs.append(mod)
        rng = np.random.RandomState()
        rng.seed(limbs)
        return rng, seed

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = np_random
    utils.seeding, utils.EzPickle, utils.colorize = seeding, EzPickle, (lambda s, *a, **k: s)
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.register = lambda **kw: None
    envs.registration = registration
    gym.Env, gym.spaces, gym.utils, gym.envs = Env, spaces, utils, envs
    gym.__path__ = []
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.utils": utils, "gym.utils.seeding": seeding,
                        "gym.envs": envs, "gym.envs.registration": registration})


# ---------------------------------------------------------------- stand-in: Box2D
class RecVec(tuple):
    x = property(lambda s: s[0])
    y = property(lambda s: s[1])


class RecBody:
    def __init__(self, kind, position, angle, fixtures):
        self.kind = kind
        self.position = RecVec((f32(position[0]), f32(position[1])))
        self.angle = f32(angle)
        self.fixtures = fixtures


class RecJoint:
    def __init__(self, d):
        self.d = d
        self.motorSpeed = 0.0

    @property
    def angle(self):
        return f32(np.float32(self.d.bodyB.angle) - np.float32(self.d.bodyA.angle))


class RecWorld:
    def __init__(self):
        self.statics, self.dynamics, self.joints = [], [], []
        self.steps = 0
        self.contactListener = None

    def CreateStaticBody(self, fixtures=None):
        shape = fixtures.shape
        b = RecBody("static", (0, 0), 0, dict(kind=shape.kind, vertices=[tuple(map(float, v)) for v in shape.vertices],
                                             friction=fixtures.friction))
        self.statics.append(b)
        return b

    def CreateDynamicBody(self, position=None, angle=0, fixtures=None):
        shape = fixtures.shape
        fx = dict(kind=shape.kind, density=fixtures.density, friction=fixtures.friction,
                  categoryBits=fixtures.categoryBits, maskBits=fixtures.maskBits)
        if shape.kind == "polygon":
            fx["box"] = (f32(shape.box[0]), f32(shape.box[1]))
        else:
            fx["radius"] = f32(shape.radius)
        b = RecBody("dynamic", position, angle, fx)
        self.dynamics.append(b)
        return b

    def CreateJoint(self, d):
        j = RecJoint(d)
        self.joints.append(j)
        return j

    def DestroyBody(self, b):
        pass

    def Step(self, dt, vi, pi):
        self.steps += 1


class _Shape:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)


def install_box2d_stub():
    B = types.ModuleType("Box2D")
    b2 = types.ModuleType("Box2D.b2")

    def polygonShape(vertices=None, box=None):
        return _Shape("polygon", vertices=list(vertices) if vertices is not None else None, box=box)

    def edgeShape(vertices=None):
        return _Shape("edge", vertices=list(vertices))

    def circleShape(radius=0, pos=(0, 0)):
        return _Shape("circle", radius=radius, pos=pos)

    class fixtureDef:
        def __init__(self, shape=None, density=0.0, friction=0.2, restitution=0.0, categoryBits=0x0001,
                     maskBits=0xFFFF):
            self.shape, self.density, self.friction, self.restitution = shape, density, friction, restitution
            self.categoryBits, self.maskBits = categoryBits, maskBits

    class revoluteJointDef:
        def __init__(self, **kw):
            self.referenceAngle = 0.0
            self.__dict__.update(kw)

    class contactListener:
        def __init__(self):
            pass

    b2.polygonShape, b2.edgeShape, b2.circleShape = polygonShape, edgeShape, circleShape
    b2.fixtureDef, b2.revoluteJointDef, b2.contactListener = fixtureDef, revoluteJointDef, contactListener
    B.b2, B.b2World, B.b2CircleShape = b2, RecWorld, circleShape
    B.b2RevoluteJoint = object
    sys.modules.update({"Box2D": B, "Box2D.b2": b2})


def install_neat_stub():
    """`NeuralNetwork/NEAT_NN.py` does `import neat` at module level (neat-python: pinned by the reference, absent here).
    Only the import has to succeed: the network capture below never builds a NEAT genome -- it hands the reference's
    `NN_enc.create()` a duck-typed `nn_g.getPhenotype().activate(x)` (Network_Encoding.py:100-101,177-178)."""
    neat = types.ModuleType("neat")

    class DefaultGenome:
        def __init__(self, key):
            self.key = key
    neat.DefaultGenome = DefaultGenome
    sys.modules["neat"] = neat


def alias_case_insensitive():
    """The reference was written on a case-insensitive file system
    (`from Encodings import abstract_encoding`, file Abstract_Encoding.py)."""
    import Encodings
    for fn, low in (("Abstract_Encoding.py", "abstract_encoding"), ("Direct_Encoding.py", "direct_encoding"),
                    ("LSystem.py", "lsystem"), ("Cellular_Encoding.py", "cellular_encoding")):
        name = "Encodings." + low
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "Encodings", fn))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        setattr(Encodings, low, mod)
        spec.loader.exec_module(mod)


# ---------------------------------------------------------------- dumps
def con_name(c):
    return None if c is None else c.name


def dump_module(m):
    d = dict(type=m.type, angle=m.angle, torque=m.torque)
    if m.type == "SIMPLE":
        d.update(width=m.width, height=m.height)
    else:
        d.update(radius=m.radius)
    return d


def dump_ctrl(c):
    return None if c is None else dict(amplitude=c.amplitude, phase=c.phase, frequency=c.frequency,
                                       offset=c.offset, i_state=c.i_state)


def dump_tree(tree):
    nodes = tree.getNodes()
    return [dict(index=n.index, parent=n.parent, type=n.type, con=con_name(n.parent_connection_coordinates),
                 module=dump_module(n.module_), controller=dump_ctrl(n.controller)) for n in nodes]


def dump_layout(env, n_ctrl_steps=20):
    w = env.world
    dyn = w.dynamics
    bodies = [dict(kind=b.fixtures["kind"], box=b.fixtures.get("box"), radius=b.fixtures.get("radius"),
                   x=b.position[0], y=b.position[1], angle=b.angle, friction=b.fixtures["friction"],
                   categoryBits=b.fixtures["categoryBits"], maskBits=b.fixtures["maskBits"]) for b in dyn]
    joints = []
    for j in w.joints:
        d = j.d
        joints.append(dict(bodyA=dyn.index(d.bodyA), bodyB=dyn.index(d.bodyB),
                           anchorA=[f32(d.localAnchorA[0]), f32(d.localAnchorA[1])],
                           anchorB=[f32(d.localAnchorB[0]), f32(d.localAnchorB[1])],
                           torque=f32(d.maxMotorTorque), lower=f32(d.lowerAngle), upper=f32(d.upperAngle),
                           enableMotor=bool(d.enableMotor), enableLimit=bool(d.enableLimit),
                           referenceAngle=f32(d.referenceAngle)))
    nodes = env.tree_morphology.nodes
    flags = [dict(expressed=bool(n.expressed), has_component=n.component is not None,
                  body=(dyn.index(n.component[0]) if n.component is not None else -1)) for n in nodes]
    # controller / PID sequence with the bodies frozen at their construction pose
    seq = []
    for _ in range(n_ctrl_steps):
        _, reward, done, _ = env.step(None)
        seq.append(dict(motorSpeed=[f32(j.motorSpeed) for j in w.joints], reward=reward, done=bool(done),
                        wod=env.wod.position))
    return dict(bodies=bodies, joints=joints, node_flags=flags, control=seq)


def main():
    sys.path.insert(0, REF)
    install_gym_stub()
    install_box2d_stub()
    import matplotlib
    matplotlib.use("Agg")
    install_neat_stub()
    alias_case_insensitive()
    from Encodings import direct_encoding as de, lsystem as ls
    from gym_rem2D.morph import simple_module, circular_module
    from gym_rem2D.envs import Modular2DEnv as M

    M.COLOR_CONTROL = False  # render-only colour lookups (Modular2DEnv.py:624-628)
    os.makedirs(OUT, exist_ok=True)

    def module_list():
        return [simple_module.Standard2D() for _ in range(4)] + [circular_module.Circular2D() for _ in range(4)]

    # ---- terrain profiles
    terr = {}
    for name, hardcore, maxp in (("default", False, 24), ("flat", False, 0), ("hardcore", True, 24)):
        M.MAX_PERTURBANCE_TERRAIN = maxp
        env = M.Modular2D()
        env.hardcore = hardcore
        env.seed(4)
        env.reset(tree=None, module_list=None)
        polys = [s.fixtures["vertices"] for s in env.world.statics if s.fixtures["kind"] == "polygon"]
        edges = [s.fixtures["vertices"] for s in env.world.statics if s.fixtures["kind"] == "edge"]
        order = [s.fixtures["kind"] for s in env.world.statics]
        assert order == ["polygon"] * len(polys) + ["edge"] * len(edges)
        terr[name] = dict(x=list(map(float, env.terrain_x)), y=list(map(float, env.terrain_y)), polys=polys,
                          n_edges=len(edges), friction=env.world.statics[-1].fixtures["friction"])
    M.MAX_PERTURBANCE_TERRAIN = 24
    with open(os.path.join(OUT, "terrain_seed4.json"), "w") as f:
        json.dump(terr, f)

    # ---- trees + layouts per encoding
    for enc, n_seeds in (("direct", 40), ("lsystem", 40)):
        cases = []
        for seed in range(n_seeds):
            random.seed(seed)
            ml = module_list()
            genome = de.DirectEncoding(ml) if enc == "direct" else ls.LSystem(ml)
            if enc == "lsystem" and seed % 2 == 1:
                # exercise mutated module sizes/angles too
                for _ in range(3):
                    genome.mutate(0.5, 0.5, 0.5)
            tree = genome.create(8)
            env = M.Modular2D()
            env.seed(4)
            env.reset(tree=tree, module_list=ml)
            cases.append(dict(seed=seed, tree=dump_tree(tree), layout=dump_layout(env)))
        with open(os.path.join(OUT, "layout_%s.json" % enc), "w") as f:
            json.dump(dict(encoding=enc, cases=cases), f)
        nb = [len(c["layout"]["bodies"]) for c in cases]
        print(enc, "bodies per creature: mean %.2f max %d" % (np.mean(nb), max(nb)))

    # ---- network encoding (BASELINE config 4's input generator): the reference's OWN NN_enc.create / update / iterate /
    # recursiveNodeGen (Network_Encoding.py:86-139,141-222) grow the tree.  Its genome class needs neat-python, so the
    # encoder is instantiated without __init__ and given what __init__ would have set (:48-84), with nn_g = this
    # repository's feed-forward CPPN (weights recorded in the fixture): the network is duck-typed (`activate(input)`),
    # everything downstream of it -- thresholds, module choice, setMorph, setControl, index / depth caps, node order --
    # is the reference's code.
    import copy
    from Encodings import Network_Encoding as ne
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from gym_rem2d_amd.encodings.network import FeedForwardCPPN
    cases = []
    for seed in range(64):
        random.seed(1000 + seed)
        ml = module_list()
        g = object.__new__(ne.NN_enc)
        g.moduleList = copy.deepcopy(ml)
        g.outputs, g.inputs = [0] * 10, []
        g.maxTreeDepth, g.maxModules = 7, 20
        g.networkType = ne.NETWORK_TYPE.CPPN
        g.nn_g = FeedForwardCPPN()            # (draws its weights from `random`, like NNEncoding.__init__ does at this point)
        for mod in g.moduleList:              # Network_Encoding.py:83-84
            mod.mutate(0.5, 0.5, 0.5)
        for _ in range(seed % 4):             # mutated genomes too (Network_Encoding.py:142-150)
            g.mutate(0.4, 0.4, 0.3)
        depth = 7 if seed % 3 else 4
        net = dict(w1=[list(map(float, r)) for r in g.nn_g.w1], a1=list(map(int, g.nn_g.a1)),
                   w2=[list(map(float, r)) for r in g.nn_g.w2])
        protos = [dict(module=dump_module(m), controller=dump_ctrl(m.controller)) for m in g.moduleList]
        tree = g.create(depth)
        env = M.Modular2D()
        env.seed(4)
        env.reset(tree=tree, module_list=g.moduleList)
        cases.append(dict(seed=1000 + seed, mutations=seed % 4, depth=depth, network=net, module_list=protos,
                          tree=dump_tree(tree), layout=dump_layout(env, n_ctrl_steps=3)))
    with open(os.path.join(OUT, "layout_network.json"), "w") as f:
        json.dump(dict(encoding="network", cases=cases), f)
    nb = [len(c["layout"]["bodies"]) for c in cases]
    print("network bodies per creature: mean %.2f max %d, nodes max %d" % (np.mean(nb), max(nb), max(len(c["tree"]) for c in cases)))

    # ---- hand-built chains (BASELINE config 2 morphology) through the reference code
    import Tree as T
    from Controller import m_controller
    chains = {}
    for site in ("top", "left", "right"):
        random.seed(123)
        tree = T.Tree([])
        for k in range(4):
            mod = simple_module.Standard2D()
            mod.width, mod.height, mod.angle = 0.5, 0.8, math.pi / 2
            c = m_controller.Controller()
            c.amplitude, c.frequency, c.phase, c.offset = 0.5, 0.1, -1 + 2 * k / 3, 0.0
            con = None if k == 0 else getattr(simple_module.Connection, site)
            tree.nodes.append(T.Node(k, k - 1, 0, con, c, module_=mod))
        env = M.Modular2D()
        env.seed(4)
        env.reset(tree=tree, module_list=[0])
        chains[site] = dict(tree=dump_tree(tree), layout=dump_layout(env))
    with open(os.path.join(OUT, "layout_chain.json"), "w") as f:
        json.dump(chains, f)

    # ---- diversity metric (DataAnalysis/AdvancedDataAnalysis.py:291-381) on seeded populations
    spec = importlib.util.spec_from_file_location("ref_advanced_data_analysis",
                                                  os.path.join(REF, "DataAnalysis", "AdvancedDataAnalysis.py"))
    ADA = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ADA)

    class _Ind:
        tree_depth = 8

        def __init__(self, genome):
            self.genome = genome

    div = {}
    for enc, n_seeds in (("lsystem", 32), ("direct", 24)):
        pop = []
        for seed in range(n_seeds):
            random.seed(seed)
            ml = module_list()
            genome = de.DirectEncoding(ml) if enc == "direct" else ls.LSystem(ml)
            if enc == "lsystem" and seed % 2 == 1:
                for _ in range(3):
                    genome.mutate(0.5, 0.5, 0.5)
            pop.append(_Ind(genome))
        positions = [[[float(v.pos[0]), float(v.pos[1])] for v in ADA.get_tree_pos(ind.genome.create(8))] for ind in pop]
        div[enc] = dict(n=n_seeds, positions=positions, diversity=[float(v) for v in ADA.tree_edit_distance(pop)])
    with open(os.path.join(OUT, "diversity.json"), "w") as f:
        json.dump(div, f)

    # ---- a population checkpoint as the reference pickles it (REM2D_main.py:311-329): reference classes,
    # reference module paths; REM2D_main itself needs deap, so its two small classes are stood in for here
    import enum
    import pickle
    main_mod = types.ModuleType("REM2D_main")

    class Encoding_Type(enum.Enum):
        DIRECT = 0
        LSYSTEM = 1
        NEURAL_NETWORK = 2
        CELLULAR_ENCODING = 3

    class Individual:
        def __init__(self):
            self.genome = None
            self.fitness = 0
    for c in (Encoding_Type, Individual):
        c.__module__, c.__qualname__ = "REM2D_main", c.__name__
        setattr(main_mod, c.__name__, c)
    sys.modules["REM2D_main"] = main_mod
    pop = []
    for enc, seeds in (("lsystem", range(6)), ("direct", range(4))):
        for seed in seeds:
            random.seed(seed)
            ml = module_list()
            ind = Individual()
            ind.ENCODING_TYPE = Encoding_Type.DIRECT if enc == "direct" else Encoding_Type.LSYSTEM
            ind.genome = de.DirectEncoding(ml) if enc == "direct" else ls.LSystem(ml)
            if enc == "lsystem" and seed % 2 == 1:
                for _ in range(3):
                    ind.genome.mutate(0.5, 0.5, 0.5)
            ind.tree_depth = 8
            ind.fitness = float(seed) + 0.5
            pop.append(ind)
    with open(os.path.join(OUT, "reference_population.pkl"), "wb") as f:
        pickle.dump(pop, f, protocol=2)
    print("wrote fixtures to", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
