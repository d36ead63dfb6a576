"""Host layer vs golden vectors captured from the reference's importable Python side
(tools/capture_golden.py -> tests/golden/*.json): terrain profile, encodings, create_robot
layout, joint anchors, controller/PID sequence.  Everything here is bit-exact."""
import copy
import json
import math
import os
import random

import numpy as np
import pytest

from conftest import oracle_terrain
from gym_rem2d_amd import Morphology, build_creature, get_module_list, make_terrain, synthetic
from gym_rem2d_amd.compiler import f32
from gym_rem2d_amd.encodings import DirectEncoding, LSystem

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("name,kw", [("default", {}), ("flat", {"flat": True}), ("hardcore", {"hardcore": True})])
def test_terrain_profile(name, kw):
    g = load("terrain_seed4.json")[name]
    t = make_terrain(4, **kw)
    assert np.array_equal(t.xs, np.array(g["x"]))
    assert np.array_equal(t.ys, np.array(g["y"]))
    assert np.array_equal(t.polys, np.array(g["polys"], dtype=np.float64).reshape(-1, 4, 2))
    assert g["n_edges"] == 199 and t.friction == g["friction"]
    if name == "flat":
        assert np.all(t.ys == 5.0)


def _genome(enc, seed):
    random.seed(seed)
    ml = get_module_list()
    g = DirectEncoding(ml) if enc == "direct" else LSystem(ml)
    if enc == "lsystem" and seed % 2 == 1:
        for _ in range(3):
            g.mutate(0.5, 0.5, 0.5)
    return g, ml


def _check_tree(nodes, gt):
    assert len(nodes) == len(gt)
    for n, gn in zip(nodes, gt):
        con = None if n.parent_connection_coordinates is None else n.parent_connection_coordinates.name
        assert (n.index, n.parent, n.type, con) == (gn["index"], gn["parent"], gn["type"], gn["con"])
        m, gm = n.module_, gn["module"]
        assert m.type == gm["type"] and m.angle == gm["angle"] and m.torque == gm["torque"]
        if m.type == "SIMPLE":
            assert (m.width, m.height) == (gm["width"], gm["height"])
        else:
            assert m.radius == gm["radius"]
        c, gc = n.controller, gn["controller"]
        assert (c.amplitude, c.phase, c.frequency, c.offset) == (gc["amplitude"], gc["phase"], gc["frequency"], gc["offset"])


def _check_layout(spec, nodes, L):
    assert len(spec.bodies) == len(L["bodies"]) and len(spec.joints) == len(L["joints"])
    for b, gb in zip(spec.bodies, L["bodies"]):
        if gb["kind"] == "polygon":
            assert b.shape == 1 and [b.hx, b.hy] == gb["box"]
        else:
            assert b.shape == 2 and b.hx == gb["radius"]
        assert (b._x, b._y, b._angle) == (gb["x"], gb["y"], gb["angle"])
        assert gb["friction"] == 0.1 and gb["categoryBits"] == 0x20 and gb["maskBits"] == 0x1
    for j, gj in zip(spec.joints, L["joints"]):
        assert (j["parent"], j["child"]) == (gj["bodyA"], gj["bodyB"])
        assert [j["ax"], j["ay"]] == gj["anchorA"] and [j["bx"], j["by"]] == gj["anchorB"]
        assert (j["torque"], j["lower"], j["upper"]) == (gj["torque"], gj["lower"], gj["upper"])
        assert gj["enableMotor"] and gj["enableLimit"] and gj["referenceAngle"] == 0.0
        assert gj["bodyB"] == L["joints"].index(gj) + 1  # joint k <-> body k+1
    for n, fl, slot in zip(nodes, L["node_flags"], spec.node_slots):
        assert bool(n.expressed) == fl["expressed"] and (n.component is not None) == fl["has_component"]
        assert slot == fl["body"]


def _check_control(spec, nodes, L):
    """Controller sweep + PID with the bodies frozen at their construction pose
    (Modular2DEnv.py:613-632), python doubles -> float32 motorSpeed."""
    ctrls = [n.controller for n in nodes if n.controller is not None and n.expressed and n.component is not None]
    angles = [f32(np.float32(spec.bodies[j["child"]]._angle) - np.float32(spec.bodies[j["parent"]]._angle))
              for j in spec.joints]
    wod = 0.0
    for k, rec in enumerate(L["control"]):
        wod += 0.04
        cv = [c.update(0) for c in ctrls]
        ms = [f32((cv[i + 1] - angles[i]) * 1.9) for i in range(len(spec.joints))]
        assert ms == rec["motorSpeed"]
        assert rec["wod"] == wod
        assert rec["reward"] == (5.0 if wod <= 5.0 else -100)


@pytest.mark.parametrize("enc", ["direct", "lsystem"])
def test_encoding_and_layout(enc):
    G = load("layout_%s.json" % enc)
    assert len(G["cases"]) == 40
    for case in G["cases"]:
        genome, ml = _genome(enc, case["seed"])
        tree = genome.create(8)
        _check_tree(tree.getNodes(), case["tree"])
        t2 = copy.deepcopy(tree)
        nodes = t2.getNodes()
        spec, comps, joints = build_creature(nodes, ml)
        _check_layout(spec, nodes, case["layout"])
        assert len(comps) == spec.n_bodies and len(joints) == len(spec.joints)
        _check_control(spec, nodes, case["layout"])


def _network_genome(case):
    """The genome of a layout_network.json case, rebuilt from its seed the way the capture built the reference's: module
    list, network weights from `random`, NN_enc.__init__'s module mutation, `mutations` genome mutations."""
    from gym_rem2d_amd.encodings import NNEncoding
    random.seed(case["seed"])
    g = NNEncoding(get_module_list())
    for _ in range(case["mutations"]):
        g.mutate(0.4, 0.4, 0.3)
    return g


def test_network_encoding_and_layout_match_the_reference():
    """BASELINE config 4's input generator, pinned to the reference: tests/golden/layout_network.json holds 64 trees grown
    by the REFERENCE's own ``NN_enc.create / update / iterate / recursiveNodeGen`` (Encodings/Network_Encoding.py:86-139,
    141-222; the encoder instantiated without its neat-python constructor and queried through the duck-typed
    ``nn_p.activate(input)`` with this repository's feed-forward net, weights recorded in the fixture) and laid out by the
    reference's ``create_robot``.  ``encodings/network.py`` must grow the same trees node by node -- thresholds, module
    choice, ``setMorph`` / ``setControl`` arguments, the index and depth caps, DFS emission with the MAX_MODULES cut --
    and the Python compiler must place every body, joint anchor and skipped node identically, bit for bit."""
    G = load("layout_network.json")
    assert len(G["cases"]) == 64
    sizes = []
    for case in G["cases"]:
        g = _network_genome(case)
        # the genome itself first: same network weights and the same (mutated) module prototypes as the capture recorded
        assert g.nn_g.w1 == case["network"]["w1"] and g.nn_g.a1 == case["network"]["a1"] and g.nn_g.w2 == case["network"]["w2"]
        for m, gm in zip(g.moduleList, case["module_list"]):
            mm, gc = gm["module"], gm["controller"]
            assert (m.type, m.angle, m.torque) == (mm["type"], mm["angle"], mm["torque"])
            assert ((m.width, m.height) == (mm["width"], mm["height"])) if m.type == "SIMPLE" else (m.radius == mm["radius"])
            c = m.controller
            assert (c.amplitude, c.phase, c.frequency, c.offset) == (gc["amplitude"], gc["phase"], gc["frequency"], gc["offset"])
        tree = g.create(case["depth"])
        _check_tree(tree.getNodes(), case["tree"])
        t2 = copy.deepcopy(tree)
        nodes = t2.getNodes()
        spec, comps, joints = build_creature(nodes, g.moduleList)
        _check_layout(spec, nodes, case["layout"])
        _check_control(spec, nodes, case["layout"])
        sizes.append(spec.n_bodies)
    assert max(sizes) >= 12 and len(set(sizes)) >= 6 and min(sizes) == 1   # the fixture population is varied


@pytest.mark.parametrize("depth", [7, 4])
def test_native_network_compiler_matches_the_reference(depth):
    """``rem2d_compile_network`` (host C++: the NN queries, tree growth, create_robot, schedule and SoA packing in one native
    pass) against the same reference-grown fixture: every body (shape, half extents, pose), joint (parent, anchors,
    torque, limits) and controller word of every case equals what the reference produced, in binary32 like pybox2d stores
    them."""
    import __graft_entry__ as ge
    ge.build()
    from gym_rem2d_amd import encode
    G = load("layout_network.json")
    cases = [c for c in G["cases"] if c["depth"] == depth]
    assert len(cases) >= 20
    genomes = [_network_genome(c) for c in cases]
    m = encode.compile_network_arrays(encode.network_genome_arrays(genomes), depth, genomes[0].maxModules, 32, n_threads=2)
    K = m.lanes
    for e, case in enumerate(cases):
        L = case["layout"]
        nb = len(L["bodies"])
        assert int(m.n_bodies[e]) == nb
        lo = e * K
        for k, gb in enumerate(L["bodies"]):
            if gb["kind"] == "polygon":
                assert m["shape"][lo + k] == 1 and [float(m["hx"][lo + k]), float(m["hy"][lo + k])] == gb["box"]
            else:
                assert m["shape"][lo + k] == 2 and float(m["hx"][lo + k]) == gb["radius"]
            assert (float(m["x"][lo + k]), float(m["y"][lo + k]), float(m["angle"][lo + k])) == (gb["x"], gb["y"], gb["angle"])
        assert (m["shape"][lo + nb:lo + K] == 0).all() and m["parent"][lo] == -1
        for gj in L["joints"]:
            k = lo + gj["bodyB"]
            assert m["parent"][k] == gj["bodyA"]
            assert [float(m["ax"][k]), float(m["ay"][k])] == gj["anchorA"] and [float(m["bx"][k]), float(m["by"][k])] == gj["anchorB"]
            assert (float(m["torque"][k]), float(m["lower"][k]), float(m["upper"][k])) == (gj["torque"], gj["lower"], gj["upper"])
        # controllers of the expressed nodes, in body order (the reference's c_values sweep, Modular2DEnv.py:620-628); the
        # root's drives no joint (c_values[i + 1] -> joint i, :631-632) and is not part of the upload format
        ctl = [n["controller"] for n, fl in zip(case["tree"], L["node_flags"]) if fl["has_component"]]
        assert len(ctl) == nb
        for k, gc in enumerate(ctl):
            if k == 0:
                continue
            assert (float(m["amp"][lo + k]), float(m["phase"][lo + k]), float(m["freq"][lo + k]), float(m["offset"][lo + k])) == \
                (gc["amplitude"], gc["phase"], gc["frequency"], gc["offset"])


@pytest.mark.parametrize("site", ["top", "left", "right"])
def test_chain_layout(site):
    G = load("layout_chain.json")[site]
    tree = synthetic.chain_tree(4, site)
    _check_tree(tree.getNodes(), G["tree"])
    t2 = copy.deepcopy(tree)
    spec, _, _ = build_creature(t2.getNodes(), [0])
    _check_layout(spec, t2.getNodes(), G["layout"])
    _check_control(spec, t2.getNodes(), G["layout"])
    if site == "top":
        for k, b in enumerate(spec.bodies):  # bodies at (5, 7 + 0.8 k)
            assert b._x == 5.0 and b._y == pytest.approx(7 + 0.8 * k, abs=1e-6) and b._angle == 0.0


def test_oracle_first_motor_speed_matches_reference(oracle, rough_terrain):
    """The oracle's controller/PID (o_sin in binary64, float32 narrowing) reproduces the
    reference's first motorSpeed for every fixture creature."""
    ot = oracle_terrain(oracle, rough_terrain)
    for enc in ("direct", "lsystem"):
        for case in load("layout_%s.json" % enc)["cases"]:
            genome, ml = _genome(enc, case["seed"])
            t2 = copy.deepcopy(genome.create(8))
            spec, _, _ = build_creature(t2.getNodes(), ml)
            if not spec.joints:
                continue
            w = oracle.World.from_morph(ot, Morphology.from_specs([spec]).as_dict(), 0)
            w.env_step()
            assert w.joints()[:, 4].tolist() == case["layout"]["control"][0]["motorSpeed"]


def test_morphology_packing():
    specs = synthetic.lsystem_specs(range(6))
    m = Morphology.from_specs(specs, 32)
    for e, s in enumerate(specs):
        lo = e * 32
        assert (m["shape"][lo:lo + 32] != 0).sum() == s.n_bodies == m.n_bodies[e]
        assert m["parent"][lo] == -1
        for k, j in enumerate(s.joints):
            assert m["parent"][lo + j["child"]] == j["parent"] and j["child"] == k + 1
            assert (m["jround"][lo + j["child"]] & 0xff) == s.rounds[k]
            assert (m["jround"][lo + j["child"]] >> 16) == s.period
    sub = m.take([4, 1])
    assert np.array_equal(sub["x"][:32], m["x"][4 * 32:5 * 32]) and np.array_equal(sub["amp"][32:], m["amp"][32:64])
    rep = Morphology.replicate(specs[0], 5)
    assert rep.n_envs == 5 and np.array_equal(rep["hx"][:rep.lanes], rep["hx"][-rep.lanes:])


def _check_pipeline(spec, iters=5):
    """Replay the modulo schedule and compare, per body, the order of operations with Box2D's
    sequential sweep: iteration by iteration, the body's joints in island order, then its contacts."""
    pairs = [(j["parent"], j["child"]) for j in spec.joints]
    P, n = spec.period, spec.n_bodies
    pos = {k: i for i, k in enumerate(spec.island_order)}
    events = []   # (tick, slot, body, label)
    for t in range(iters):
        for k, (a, b) in enumerate(pairs):
            for x in (a, b):
                events.append((spec.rounds[k] + t * P, 0, x, ("j", t, pos[k])))
        for b in range(n):
            events.append((spec.offC[b] + t * P, 1, b, ("c", t, 0)))
    for b in range(n):
        mine = sorted(e for e in events if e[2] == b)
        # no two joints of one body in the same slot
        slots = [(e[0], e[1]) for e in mine]
        assert len(set(slots)) == len(slots)
        got = [e[3] for e in mine]
        want = sorted(got, key=lambda l: (l[1], 0 if l[0] == "j" else 1, l[2]))
        assert got == want, (b, got[:8], want[:8])
    assert max(spec.offC, default=0) < 256 and P < 256


def test_pipeline_schedule_keeps_sequential_order():
    from gym_rem2d_amd import synthetic
    specs = synthetic.lsystem_specs(list(range(300))) + synthetic.direct_specs(list(range(100)))
    specs += synthetic.cppn_specs(list(range(60)))
    assert any(s.period >= 4 for s in specs)
    for s in specs:
        _check_pipeline(s)


def test_reference_checkpoint_loads_and_expresses_identically():
    """tests/golden/reference_population.pkl was pickled from the reference's own classes
    (tools/capture_golden.py; module paths Encodings.*, gym_rem2D.morph.*, Controller.*, REM2D_main).
    compat.load_reference_pickle maps them onto this package; the loaded genomes must express to the layouts
    the reference produced for the same seeds, and survive mutation + the native compiler."""
    from gym_rem2d_amd.compat import Encoding_Type, load_reference_pickle
    from gym_rem2d_amd.ea import Individual
    pop = load_reference_pickle(os.path.join(GOLD, "reference_population.pkl"))
    assert len(pop) == 10 and all(isinstance(i, Individual) for i in pop)
    assert [i.fitness for i in pop] == [s + 0.5 for s in range(6)] + [s + 0.5 for s in range(4)]
    assert pop[0].ENCODING_TYPE is Encoding_Type.LSYSTEM and pop[-1].ENCODING_TYPE is Encoding_Type.DIRECT
    for enc, inds in (("lsystem", pop[:6]), ("direct", pop[6:])):
        cases = load("layout_%s.json" % enc)["cases"]
        for seed, ind in enumerate(inds):
            tree = ind.genome.create(ind.tree_depth)
            _check_tree(tree.getNodes(), cases[seed]["tree"])
            t2 = copy.deepcopy(tree)
            spec, _, _ = build_creature(t2.getNodes(), ind.genome.moduleList)
            _check_layout(spec, t2.getNodes(), cases[seed]["layout"])
    # loaded genomes are live objects of this package: they mutate and feed the array / native path
    random.seed(99)
    pop[0].mutate(0.5, 0.5, 0.5)
    from gym_rem2d_amd.encode import lsystem_genome_arrays
    a = lsystem_genome_arrays([i.genome for i in pop[:6]])
    assert a["mod_shape"].shape == (6, 8) and a["rule_n"].max() <= 3
    with pytest.raises(Exception, match="no counterpart"):   # a genome type this build has no class for
        load_reference_pickle(b"cEncodings.Network_Encoding\nNN_enc\n.")


def test_checkpoints_are_written_in_the_reference_format():
    """compat.dump_reference_pickle: a checkpoint read from the reference and written back names exactly the classes
    the reference's own pickle names (REM2D_main.py:311-329), and reads back to the same genomes.  (That such files
    really load inside the reference's classes and express to the same robots is checked in the build container by
    tools/check_reference_roundtrip.py, which imports /root/reference.)"""
    import pickle
    import pickletools
    from gym_rem2d_amd import compat, ea
    path = os.path.join(GOLD, "reference_population.pkl")
    pop = compat.load_reference_pickle(path)
    blob = compat.dumps_reference_pickle(pop)

    def globs(b):
        return sorted({a for op, a, _ in pickletools.genops(b) if op.name == "GLOBAL"})
    assert globs(blob) == globs(open(path, "rb").read())
    again = compat.load_reference_pickle(blob)
    assert [type(i.genome).__name__ for i in again] == [type(i.genome).__name__ for i in pop]
    for a, b in zip(pop, again):
        ta, tb = a.genome.create(8).getNodes(), b.genome.create(8).getNodes()
        assert [(n.index, n.parent, n.type) for n in ta] == [(n.index, n.parent, n.type) for n in tb]
        assert a.fitness == b.fitness
    # a genome without a counterpart in the reference is refused, not written under this package's own path
    random.seed(0)
    with pytest.raises(pickle.PicklingError, match="no counterpart"):
        compat.dumps_reference_pickle([ea.Individual.random(encoding="cppn")])


def test_gym_make_registration_and_time_limit():
    """``gym.make('Modular2DLocomotion-v0')`` (gym_rem2D/__init__.py:5-7, REM2D_main.getEnv :57-67): the id resolves to
    the env facade wrapped in a 4800-step TimeLimit.  (No GPU needed: the facade only touches it at reset(tree).)"""
    import gym_rem2d_amd
    from gym_rem2d_amd import gymshim
    from gym_rem2d_amd.env import Modular2D
    env = gym_rem2d_amd.make("Modular2DLocomotion-v0")
    assert isinstance(env, gymshim.TimeLimit) and isinstance(env.unwrapped, Modular2D)
    assert env._max_episode_steps == 4800
    assert env.seed(4) == [4] and env.action_space.shape == (4,) and env.observation_space.shape == (24,)
    assert env.reset() is None                      # Modular2D.reset() without a tree builds nothing (Modular2DEnv.py:565-598)
    with pytest.raises(Exception, match="no tree_morphology"):
        env.step(None)
    with pytest.raises(KeyError):
        gym_rem2d_amd.make("NoSuchEnv-v0")

    class Stub:
        def reset(self):
            return 0

        def step(self, a):
            return 0, 1.0, 0, 0
    tl = gymshim.TimeLimit(Stub(), 3)
    with pytest.raises(AssertionError):
        tl.step(None)
    tl.reset()
    outs = [tl.step(None) for _ in range(3)]
    assert [o[2] for o in outs] == [0, 0, True] and outs[2][3] == {"TimeLimit.truncated": True} and outs[0][3] == 0


def test_morphology_concat_take_and_uniform_population_detection():
    """Host helpers behind BatchedModular2D.compact and the launch-shape rules: concat is the inverse of take, and a
    fixed-morphology population (every creature the same tree and schedule) is told from a mixed one."""
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology
    from gym_rem2d_amd.env import _uniform
    specs = synthetic.lsystem_specs(range(24))
    m = Morphology.from_specs(specs, 16)
    a, b = m.take(np.arange(0, 10)), m.take(np.arange(10, 24))
    back = Morphology.concat([a, b])
    assert back.n_envs == m.n_envs and back.lanes == m.lanes
    for k in m.arrays:
        assert np.array_equal(back.arrays[k], m.arrays[k]), k
    assert np.array_equal(back.n_bodies, m.n_bodies)
    with pytest.raises(ValueError):
        Morphology.concat([a, Morphology.from_specs(specs[:2], 32)])
    assert not _uniform(m)
    assert _uniform(synthetic.chain_population(50, 8, "left"))
    assert _uniform(Morphology.replicate(specs[3], 7, 16))


def test_position_error_threshold_constant():
    """rem2d_position.h compares the SQUARED joint position error with POS_SLOP_SQ_MAX instead of taking the square root:
    the literal must be the largest binary32 y with sqrtf(y) <= b2_linearSlop (0.005f), so that the test decides exactly
    like b2RevoluteJoint::SolvePositionConstraints' `positionError <= b2_linearSlop` (sqrt is monotone)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "gym_rem2d_amd", "csrc", "rem2d_position.h")).read()
    lit = re.search(r"#define POS_SLOP_SQ_MAX (0x[0-9a-fp.\-]+)f", src).group(1)
    y = np.float32(float.fromhex(lit))
    slop = np.float32(0.005)
    assert float(y) == float.fromhex(lit)                                    # the literal is a binary32 value
    assert np.sqrt(y, dtype=np.float32) <= slop
    assert np.sqrt(np.nextafter(y, np.float32(1), dtype=np.float32), dtype=np.float32) > slop
    rng = np.random.default_rng(0)
    x = (rng.random(200000, dtype=np.float32) * np.float32(6e-5)).astype(np.float32)
    assert np.array_equal(np.sqrt(x, dtype=np.float32) <= slop, x <= y)
