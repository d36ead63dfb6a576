"""The oracle held to the REAL Box2D -- dormant in this image.

`tests/golden/box2d_transitions.json` is written by `tools/capture_box2d_golden.py` on a machine where `import Box2D` works (the
reference's pin, `/root/reference/requirements.txt:1`: Box2D==2.3.10) by running the reference's own `reset()` / `step()`
(`Modular2DEnv.py:565-653`, the step at `:634`) over the real engine.  Neither the build container nor the GPU box has the wheel,
so the file is absent and `test_oracle_matches_real_box2d_transitions` SKIPS: until someone runs the capture, DESIGN.md 2's
"parity unpinned" stands.  `test_pipeline_selfcheck` runs the same comparison code on a fixture of the same schema that the
capture tool writes from the oracle itself (a temporary file): it proves that capture -> fixture -> re-synchronise -> step ->
compare works end to end, and nothing about Box2D.

Protocol: (i) TRANSITION parity -- before every step the oracle is re-synchronised to the recording (poses,
velocities, awake flags, joint impulses, warm-start impulses: everything pybox2d exposes), stepped once and compared with the next
record, so differences cannot accumulate; integers (contact lists: static index, touching, manifold type, point count, feature
keys; awake; done) exactly, floats within |d| <= 1e-5 + 1e-4 |x| (impulses 1e-4 + 1e-3 |x|).  `==` is impossible by
construction: b2Rot::Set calls libm sinf / cosf in the wheel and the documented "rem2d trig" here.  A transition
whose contact-list structure differs BEFORE the step (an earlier difference in an unobservable -- sleep timer, fat AABB -- or a
genuine divergence) is counted as unaligned, not compared, and at most 5 % may be.  (ii) free-flight prefix: from reset up to the
first touching contact, WITHOUT re-synchronisation, poses within 1e-4.
"""
import json
import os
import sys

import numpy as np
import pytest

from conftest import oracle_terrain

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "box2d_transitions.json")
sys.path.insert(0, os.path.join(ROOT, "tools"))

POSE_TOL = (1e-5, 1e-4)      # |d| <= a + r |x|   
IMPULSE_TOL = (1e-4, 1e-3)
PREFIX_TOL = 1e-4            # (ii)
MAX_UNALIGNED = 0.05


def _structure(contacts):
    """the integer part of a record's contact lists: per body [(static, touching, type, pointCount, key0, key1)]"""
    return [[tuple(int(v) for v in row[:6]) for row in body] for body in contacts]


def _excess(got, want, tol):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if got.size == 0:
        return 0.0
    return float(np.max(np.abs(got - want) - (tol[0] + tol[1] * np.abs(want))))


def hold_oracle_to(doc, O, terrain):
    """Run the transition protocol over every case of a fixture document; returns the report (no assertion here)."""
    import capture_box2d_golden as cap
    ot = oracle_terrain(O, terrain)
    rep = dict(transitions=0, unaligned=0, int_mismatches=0, worst_pose=-1.0, worst_impulse=-1.0, prefix_steps=0,
               worst_prefix=0.0, reward_mismatches=0, first_problem=None)

    def problem(what, case, k):
        if rep["first_problem"] is None:
            rep["first_problem"] = "%s: %s seed %d, transition %d" % (what, case["encoding"], case["seed"], k)

    for case in doc["cases"]:
        m = cap.creature_morphology(case["encoding"], case["seed"])
        w = O.World.from_morph(ot, m.as_dict(), 0, flags=O.FLAG_CONTINUOUS)   # b2World(): continuousPhysics on
        free = O.World.from_morph(ot, m.as_dict(), 0, flags=O.FLAG_CONTINUOUS)
        assert w.n_bodies == case["n_bodies"], "the creature itself differs (layout fixtures: tests/test_host_golden.py)"
        parent = m.arrays["parent"].reshape(m.n_envs, m.lanes)[0]
        assert [[int(parent[k]), k] for k in range(1, w.n_bodies)] == case["joint_bodies"]
        states = case["states"]
        # the reset state: float32 poses on both sides, velocities zero
        s0 = cap.oracle_record(w)
        assert _excess(s0["bodies"], states[0]["bodies"], (0.0, 0.0)) <= 0.0, "reset pose differs"
        in_prefix = True
        for k in range(len(states) - 1):
            rec, nxt = states[k], states[k + 1]
            # (ii) the free-flight prefix, never re-synchronised
            if in_prefix:
                if any(row[1] for body in rec["contacts"] for row in body):
                    in_prefix = False
                else:
                    free.env_step()
                    d = np.abs(np.asarray(cap.oracle_record(free)["bodies"])[:, :3] - np.asarray(nxt["bodies"])[:, :3]).max()
                    rep["prefix_steps"] += 1
                    rep["worst_prefix"] = max(rep["worst_prefix"], float(d))
            # (i) transition: align, re-synchronise, step, compare
            rep["transitions"] += 1
            mine = cap.oracle_record(w)
            aligned = _structure(mine["contacts"]) == _structure(rec["contacts"])
            for b, row in enumerate(rec["bodies"]):
                w.set_body_state(b, *row[:6], awake=row[6])
            for j, row in enumerate(rec["joints"]):
                w.set_joint_impulses(j, *row[:4])
            if aligned:
                for b, body in enumerate(rec["contacts"]):
                    for c, row in enumerate(body):
                        w.set_contact_impulses(b, c, *row[6:10])
            r, d = w.env_step()
            if not aligned:
                rep["unaligned"] += 1
                problem("contact lists differ before the step", case, k)
                continue
            got = cap.oracle_record(w)
            if _structure(got["contacts"]) != _structure(nxt["contacts"]) or \
                    [row[6] for row in got["bodies"]] != [int(row[6]) for row in nxt["bodies"]] or bool(d) != bool(case["done"][k]):
                rep["int_mismatches"] += 1
                problem("integer state differs after the step", case, k)
                continue
            pose = _excess([row[:6] for row in got["bodies"]], [row[:6] for row in nxt["bodies"]], POSE_TOL)
            imp = max(_excess(got["joints"], nxt["joints"], IMPULSE_TOL),
                      _excess([row[6:10] for body in got["contacts"] for row in body],
                              [row[6:10] for body in nxt["contacts"] for row in body], IMPULSE_TOL))
            rep["worst_pose"], rep["worst_impulse"] = max(rep["worst_pose"], pose), max(rep["worst_impulse"], imp)
            if pose > 0.0 or imp > 0.0:
                problem("floats beyond tolerance (pose excess %.3g, impulse excess %.3g)" % (pose, imp), case, k)
            if abs(r - case["reward"][k]) > POSE_TOL[0] + POSE_TOL[1] * abs(case["reward"][k]):
                rep["reward_mismatches"] += 1
                problem("reward differs", case, k)
    return rep


def _verdict(rep):
    assert rep["transitions"] > 0
    assert rep["unaligned"] <= MAX_UNALIGNED * rep["transitions"], rep
    assert rep["int_mismatches"] == 0, rep
    assert rep["worst_pose"] <= 0.0 and rep["worst_impulse"] <= 0.0, rep
    assert rep["reward_mismatches"] == 0, rep
    assert rep["worst_prefix"] <= PREFIX_TOL, rep


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="tests/golden/box2d_transitions.json absent: no machine with the Box2D wheel has run "
                                                        "tools/capture_box2d_golden.py yet (parity with the real engine stays unpinned)")
def test_oracle_matches_real_box2d_transitions(oracle, rough_terrain):
    doc = json.load(open(FIXTURE))
    assert doc["engine"]["name"] == "pybox2d", "only a capture from the real engine may live at tests/golden/box2d_transitions.json"
    assert (doc["dt"], doc["vel_iters"], doc["pos_iters"]) == (1.0 / 50, 180, 60)
    rep = hold_oracle_to(doc, oracle, rough_terrain)
    print("box2d pin:", json.dumps(rep))
    _verdict(rep)


def test_pipeline_selfcheck(oracle, rough_terrain, tmp_path):
    """capture (--engine oracle, a temporary file) -> fixture -> hold_oracle_to: every transition aligned and identical, because both
    sides are the oracle.  Pins nothing; proves the dormant test above can run the day the real fixture exists."""
    import capture_box2d_golden as cap
    out = str(tmp_path / "selfcheck.json")
    assert cap.main(["--engine", "oracle", "--direct", "3", "--lsystem", "3", "--steps", "70", "--out", out]) == 0
    doc = json.load(open(out))
    assert doc["engine"]["name"] == "oracle-selfcheck" and len(doc["cases"]) == 6
    rep = hold_oracle_to(doc, oracle, rough_terrain)
    assert rep["transitions"] == 6 * 70 and rep["unaligned"] == 0 and rep["int_mismatches"] == 0, rep
    assert rep["worst_pose"] <= 0.0 and rep["worst_impulse"] <= 0.0 and rep["worst_prefix"] == 0.0 and rep["prefix_steps"] > 0, rep
    _verdict(rep)
    # the comparison is not vacuous: a recording that is off by more than the tolerance in ONE velocity is refused
    doc["cases"][0]["states"][40]["bodies"][0][3] += 1e-2
    bad = hold_oracle_to(doc, oracle, rough_terrain)
    assert bad["worst_pose"] > 0.0 and "floats beyond tolerance" in bad["first_problem"]
    # and the real fixture's slot refuses anything but the real engine; without the wheel the capture tool writes nothing
    assert cap.main(["--engine", "oracle", "--out", cap.OUT_DEFAULT]) == 2
    try:
        import Box2D  # noqa: F401
    except ImportError:
        assert cap.main(["--engine", "box2d", "--out", str(tmp_path / "never.json")]) == 2
        assert not os.path.exists(str(tmp_path / "never.json"))
