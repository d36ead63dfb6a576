#!/usr/bin/env python3
"""The dormant pin of the engine arithmetic: transition fixtures from the REAL Box2D, wherever it can be imported.

`Modular2DEnv.py:634` (`self.world.Step(1.0/FPS, 6*30, 2*30)`) enters the third-party wheel `Box2D==2.3.10`
(`/root/reference/requirements.txt:1`), which exists on neither the build container nor the GPU box -- so in THIS image the script
stops at `import Box2D` and writes nothing, and `tests/test_box2d_pin.py` skips.  On any machine that has the wheel (and a checkout
of the reference), one command turns the oracle's "parity unpinned" into a measured statement:

    pip install Box2D==2.3.10            # (the reference's own pin)
    python tools/capture_box2d_golden.py --reference /path/to/gym_rem2D/ModularER_2D
    python -m pytest tests/test_box2d_pin.py -q

What it does: imports the reference IN PLACE (nothing is copied; `gym` is replaced by the same stand-in tools/capture_golden.py
uses when the package is missing, `neat` likewise), builds the creatures of `tests/golden/layout_{direct,lsystem}.json` from their
seeds exactly as tools/capture_golden.py does, runs the reference's own `reset()` + T x `step()` over real Box2D, and after
reset and after every step records everything pybox2d lets a script read of the b2World -- per body pose, velocity, awake flag;
per joint the accumulated impulses (`GetReactionForce(1)`, `GetReactionTorque(1)`, `GetMotorTorque(1)`) and `motorSpeed`; per
body its contact list in list order (`body.contacts`) with the static body's index, `touching`, manifold type, point count,
feature keys and warm-start impulses; the wall of death; reward / done.  Consecutive records are TRANSITIONS (SURVEY 8c protocol
(i): full state in -> one step out); `tests/test_box2d_pin.py` re-synchronises the oracle to record k before stepping it and
compares with record k + 1.

`--engine oracle` writes the same schema from the repository's own CPU oracle -- NOT a pin of anything (the oracle against
itself), only the way the capture -> fixture -> test pipeline is exercised in an image without the wheel
(`tests/test_box2d_pin.py::test_pipeline_selfcheck` does so into a temporary file; such a file must never be committed as
`tests/golden/box2d_transitions.json`, and the test refuses one there by its `engine.name`).
"""
import argparse
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DEFAULT = os.path.join(ROOT, "tests", "golden", "box2d_transitions.json")
DT, VEL_ITERS, POS_ITERS = 1.0 / 50, 6 * 30, 2 * 30   # Modular2DEnv.py:634


def case_list(n_direct, n_lsystem):
    return [("direct", s) for s in range(n_direct)] + [("lsystem", s) for s in range(n_lsystem)]


# ------------------------------------------------------------------------------------------------ real Box2D
def capture_box2d(reference, cases, steps):
    import Box2D   # noqa: F401  -- ImportError here is the normal outcome in the build image (main() reports it)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import capture_golden as cg   # the stand-ins for gym / neat and the case-insensitive aliases (NOT its Box2D recorder)
    cg.REF = reference
    sys.path.insert(0, reference)
    try:
        import gym  # noqa: F401
    except ImportError:
        cg.install_gym_stub()
    try:
        import neat  # noqa: F401
    except ImportError:
        cg.install_neat_stub()
    import matplotlib
    matplotlib.use("Agg")
    cg.alias_case_insensitive()
    from Encodings import direct_encoding as de, lsystem as ls
    from gym_rem2D.morph import simple_module, circular_module
    from gym_rem2D.envs import Modular2DEnv as M
    M.COLOR_CONTROL = False   # render-only colour lookups (Modular2DEnv.py:624-628)

    def record(env):
        comps, joints = env.robot.components, env.robot.joints
        bodies = [[b.position[0], b.position[1], b.angle, b.linearVelocity[0], b.linearVelocity[1], b.angularVelocity, int(b.awake)]
                  for b in comps]
        js = []
        for j in joints:
            f = j.GetReactionForce(1.0)   # inv_dt = 1: the accumulated impulse itself (b2RevoluteJoint::GetReactionForce)
            js.append([f[0], f[1], j.GetReactionTorque(1.0), j.GetMotorTorque(1.0), j.motorSpeed])
        contacts = []
        for b in comps:
            rows = []
            for ce in b.contacts:   # m_contactList of the body, head first
                c = ce.contact
                other = ce.other
                m = c.manifold
                pts = list(m.points)[:m.pointCount]
                key = [int(p.id.key) for p in pts] + [0, 0]
                ni = [float(p.normalImpulse) for p in pts] + [0.0, 0.0]
                ti = [float(p.tangentImpulse) for p in pts] + [0.0, 0.0]
                mtype = getattr(m, "type_", None)          # (pybox2d names b2Manifold::type `type_`; older SWIG layers `type`)
                if mtype is None:
                    mtype = getattr(m, "type")
                rows.append([int(other.userData["static"]), int(c.touching), int(mtype), int(m.pointCount), key[0], key[1],
                             ni[0], ni[1], ti[0], ti[1]])
            contacts.append(rows)
        return {"bodies": bodies, "joints": js, "contacts": contacts, "wod": env.wod.position}

    out = []
    for enc, seed in cases:
        random.seed(seed)
        ml = [simple_module.Standard2D() for _ in range(4)] + [circular_module.Circular2D() for _ in range(4)]
        genome = de.DirectEncoding(ml) if enc == "direct" else ls.LSystem(ml)
        if enc == "lsystem" and seed % 2 == 1:
            for _ in range(3):
                genome.mutate(0.5, 0.5, 0.5)
        tree = genome.create(8)
        env = M.Modular2D()
        env.seed(4)
        env.reset(tree=tree, module_list=ml)
        for i, t in enumerate(env.terrain):     # creation order == the oracle's static index (hardcore boxes, then edges)
            t.userData = {"static": i}
        comps = env.robot.components
        joints = [[next(k for k, b in enumerate(comps) if b == j.bodyA), next(k for k, b in enumerate(comps) if b == j.bodyB)]
                  for j in env.robot.joints]
        states, rewards, dones = [record(env)], [], []
        for _ in range(steps):
            _, reward, done, _ = env.step(None)
            states.append(record(env))
            rewards.append(float(reward))
            dones.append(bool(done))
        out.append({"encoding": enc, "seed": seed, "n_bodies": len(comps), "joint_bodies": joints, "states": states,
                    "reward": rewards, "done": dones})
    import Box2D as B
    return {"name": "pybox2d", "version": getattr(B, "__version__", "unknown")}, out


# ------------------------------------------------------------------------------------------------ the oracle (pipeline self-check only)
def creature_morphology(enc, seed):
    """The same creature through THIS repository's restatement of the encodings and of create_robot (held to the reference's
    layouts by tests/test_host_golden.py)."""
    from gym_rem2d_amd import Morphology, synthetic
    if enc == "direct":
        specs = synthetic.direct_specs([seed])
    else:
        specs = synthetic.lsystem_specs([seed], max_modules=20, mutate_odd=True)   # (LSystem.py:139: the default the layout fixtures use)
    return Morphology.from_specs(specs, 32)


def oracle_record(w):
    b = w.bodies()
    j = w.joints()
    contacts = []
    for k in range(w.n_bodies):
        ci, cf = w.contacts(k)
        contacts.append([[int(r[0]), int(r[3]), int(r[2]), int(r[1]), int(r[4]) if r[1] > 0 else 0, int(r[5]) if r[1] > 1 else 0,
                          float(f[0]) if r[1] > 0 else 0.0, float(f[1]) if r[1] > 1 else 0.0,
                          float(f[2]) if r[1] > 0 else 0.0, float(f[3]) if r[1] > 1 else 0.0] for r, f in zip(ci, cf)])
    return {"bodies": [[float(v) for v in r[:6]] + [int(r[7])] for r in b],
            "joints": [[float(v) for v in r[:5]] for r in j], "contacts": contacts, "wod": float(w.wod)}


def capture_oracle(cases, steps):
    from gym_rem2d_amd import make_terrain
    from oracle import oracle as O
    O.build()
    terrain = make_terrain(4)
    xs, ys, polys = terrain.f32()
    ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
    out = []
    for enc, seed in cases:
        m = creature_morphology(enc, seed)
        w = O.World.from_morph(ot, m.as_dict(), 0, flags=O.FLAG_CONTINUOUS)
        parent = m.arrays["parent"].reshape(m.n_envs, m.lanes)[0]
        states, rewards, dones = [oracle_record(w)], [], []
        for _ in range(steps):
            r, d = w.env_step()
            states.append(oracle_record(w))
            rewards.append(float(r))
            dones.append(bool(d))
        out.append({"encoding": enc, "seed": seed, "n_bodies": int(w.n_bodies),
                    "joint_bodies": [[int(parent[k]), k] for k in range(1, int(w.n_bodies))], "states": states,
                    "reward": rewards, "done": dones})
    return {"name": "oracle-selfcheck", "version": "rem2d_oracle.c"}, out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", default="/root/reference/ModularER_2D")
    ap.add_argument("--engine", choices=("box2d", "oracle"), default="box2d")
    ap.add_argument("--direct", type=int, default=12, help="direct-encoding creatures (seeds 0..N-1 of layout_direct.json)")
    ap.add_argument("--lsystem", type=int, default=12, help="L-system creatures (seeds 0..N-1 of layout_lsystem.json)")
    ap.add_argument("--steps", type=int, default=120, help="steps per creature: the fall, the landing and the first strides")
    ap.add_argument("--out", default=OUT_DEFAULT)
    args = ap.parse_args(argv)
    sys.path.insert(0, ROOT)
    cases = case_list(args.direct, args.lsystem)
    if args.engine == "box2d":
        try:
            engine, out = capture_box2d(args.reference, cases, args.steps)
        except ImportError as e:
            print("capture_box2d_golden: %s -- the Box2D wheel (requirements.txt:1 of the reference: Box2D==2.3.10) is not installed "
                  "here; nothing captured, tests/test_box2d_pin.py stays skipped" % e, file=sys.stderr)
            return 2
    else:
        if os.path.abspath(args.out) == os.path.abspath(OUT_DEFAULT):
            print("capture_box2d_golden: --engine oracle is a pipeline self-check and must not be written to %s" % OUT_DEFAULT,
                  file=sys.stderr)
            return 2
        engine, out = capture_oracle(cases, args.steps)
    doc = {"schema": 1, "engine": engine, "dt": DT, "vel_iters": VEL_ITERS, "pos_iters": POS_ITERS, "terrain_seed": 4,
           "record": {"bodies": "x y angle vx vy w awake", "joints": "impulse.x impulse.y impulse.z motorImpulse motorSpeed",
                      "contacts": "per body, list order: static touching manifold.type pointCount key0 key1 nImp0 nImp1 tImp0 tImp1"},
           "cases": out}
    with open(args.out, "w") as f:
        json.dump(doc, f)
    print("wrote %s: %d creatures x %d transitions (%s %s)" % (args.out, len(out), args.steps, engine["name"], engine["version"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
