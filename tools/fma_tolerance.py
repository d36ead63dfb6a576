#!/usr/bin/env python3
"""Validation of the TOLERANCE-MODE build (gym_rem2d_amd/librem2d_fma.so: the same source with -ffp-contract=fast) against the
strict, bit-exact build -- SURVEY.md 8c's protocol (i), transition parity: identical full state in, one env-step out.

Trajectories are chaotic after contact, so the two builds are never compared along their own trajectories: world A (strict)
runs the trajectory; before every step its whole state arena is copied into world B (fma), both advance ONE step, the outputs
are compared, and A's output is the next input of both.  Per transition and body:
  * integer state (awake, limit state, pair count, per pair slot: edge, manifold point count / type, the two feature keys;
    per creature: done, steps, TOI events) must be equal;
  * float state (pose, velocity, sleep time, joint and contact impulses) within  |d| <= ATOL + RTOL |x|.
A fused multiply-add changes a result by an ulp or two, and an ulp can flip a comparison (a limit reached a step earlier, a
contact point dropped, a TOI root found one iteration sooner): such transitions exist, are rare, and are COUNTED -- the test
states the fractions it accepts.  Run as a script on the GPU box it prints the table; tests/test_parity_gpu.py asserts it.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ATOL, RTOL = 1e-5, 1e-4
INT_LANE = ("awake", "jlimit", "ccount")
INT_SLOT = ("cedge", "cinfo", "ckey0", "ckey1")
INT_ENV = ("done", "everdone", "steps", "toievents", "err")
F_LANE = ("px", "py", "ang", "vx", "vy", "w", "sleept", "jimpx", "jimpy", "jimpz", "jmotorimp", "jmotorspeed")
F_SLOT = ("cn0", "cn1", "ct0", "ct1")


def transition_parity(morph, terrain, steps, flags, device="cuda:0"):
    """-> dict of counts over steps x creatures transitions (see the module docstring)."""
    import torch
    from gym_rem2d_amd.world import BatchedWorld
    A = BatchedWorld(morph.n_envs, morph.lanes, flags=flags, device=device)
    B = BatchedWorld(morph.n_envs, morph.lanes, flags=flags, device=device, wide="fma")
    assert A.arena.numel() == B.arena.numel()
    for w in (A, B):
        w.set_terrain(terrain)
        w.reset(morph)
    n, K = morph.n_envs, morph.lanes
    active = torch.as_tensor(morph.arrays["shape"].reshape(n, K) != 0, device=A.device)
    out = {"transitions": 0, "body_transitions": 0, "int_mismatch_creatures": 0, "float_out_of_tol_creatures": 0,
           "toi_event_transitions": 0, "max_abs_dev": 0.0, "max_abs_dev_in_tol_creatures": 0.0, "errors": 0,
           "bit_identical_creatures": 0, "by_field": {f: [0, 0.0] for f in F_LANE + F_SLOT},
           "pose_velocity_out_of_tol_creatures": 0}
    for _ in range(steps):
        B.arena.copy_(A.arena)          # identical full state in (the arena IS the state between two steps)
        toi0 = A.view("toievents").clone()
        A.step(1)
        B.step(1)
        torch.cuda.synchronize()
        bad_int = torch.zeros(n, dtype=torch.bool, device=A.device)
        for f in INT_LANE:
            bad_int |= ((A.view(f) != B.view(f)) & active).any(dim=1)
        cc = A.view("ccount")
        slot_live = (torch.arange(A.contact_slots, device=A.device).view(-1, 1, 1) < cc.unsqueeze(0)) & active.unsqueeze(0)  # [slots, n, K]
        for f in INT_SLOT:
            va, vb = A.view(f), B.view(f)
            bad_int |= ((va != vb) & slot_live).any(dim=2).any(dim=0)
        for f in INT_ENV:
            bad_int |= A.view(f) != B.view(f)
        bad_f = torch.zeros(n, dtype=torch.bool, device=A.device)
        bad_pv = torch.zeros(n, dtype=torch.bool, device=A.device)
        exact = ~bad_int
        dev = torch.zeros(n, device=A.device)
        for f in F_LANE:
            va, vb = A.view(f), B.view(f)
            d = ((va - vb).abs() * active)
            d = torch.nan_to_num(d, nan=float("inf"))
            bf = (d > ATOL + RTOL * va.abs()).any(dim=1)
            bad_f |= bf
            if f in ("px", "py", "ang", "vx", "vy", "w"):
                bad_pv |= bf
            out["by_field"][f][0] += int((bf & ~bad_int).sum())
            out["by_field"][f][1] = max(out["by_field"][f][1], float(d[~bad_int].max()) if bool((~bad_int).any()) else 0.0)
            exact &= (d == 0).all(dim=1)
            dev = torch.maximum(dev, d.max(dim=1).values)
        for f in F_SLOT:
            va, vb = A.view(f), B.view(f)
            d = (va - vb).abs() * slot_live
            d = torch.nan_to_num(d, nan=float("inf"))
            bf = (d > ATOL + RTOL * va.abs()).any(dim=2).any(dim=0)
            bad_f |= bf
            out["by_field"][f][0] += int((bf & ~bad_int).sum())
            out["by_field"][f][1] = max(out["by_field"][f][1], float(d[:, ~bad_int].max()) if bool((~bad_int).any()) else 0.0)
            exact &= (d == 0).all(dim=2).all(dim=0)
            dev = torch.maximum(dev, d.amax(dim=(0, 2)))
        out["transitions"] += n
        out["body_transitions"] += int(active.sum())
        out["int_mismatch_creatures"] += int(bad_int.sum())
        out["float_out_of_tol_creatures"] += int((bad_f & ~bad_int).sum())
        out["pose_velocity_out_of_tol_creatures"] += int((bad_pv & ~bad_int).sum())
        out["bit_identical_creatures"] += int(exact.sum())
        out["toi_event_transitions"] += int((A.view("toievents") != toi0).sum())
        out["max_abs_dev"] = max(out["max_abs_dev"], float(dev.max()))
        ok = ~(bad_f | bad_int)
        if bool(ok.any()):
            out["max_abs_dev_in_tol_creatures"] = max(out["max_abs_dev_in_tol_creatures"], float(dev[ok].max()))
    out["errors"] = int(A.view("err").max()) | int(B.view("err").max())
    A.close()
    B.close()
    return out


def default_population(n=1024):
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology
    specs = [s for s in synthetic.lsystem_specs(range(n + n // 2), mutate_odd=True) if s.n_bodies <= 16][:n]
    return Morphology.from_specs(specs, 16)


if __name__ == "__main__":
    import argparse
    from gym_rem2d_amd import _lib, make_terrain
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=150)
    a = ap.parse_args()
    m = default_population(a.n)
    for name, terrain in (("rough", make_terrain(4)), ("flat", make_terrain(4, flat=True)), ("hardcore", make_terrain(4, hardcore=True))):
        r = transition_parity(m, terrain, a.steps, _lib.FLAG_CONTINUOUS)
        t = r["transitions"]
        print("%-8s %d transitions (%d body transitions, %d with a TOI event): bit-identical %.4f %%, integer state differs %d "
              "(%.5f %%), floats beyond %.0e + %.0e |x| %d (%.5f %%), largest deviation %.3g (%.3g among the accepted), err %d"
              % (name, t, r["body_transitions"], r["toi_event_transitions"], 100.0 * r["bit_identical_creatures"] / t,
                 r["int_mismatch_creatures"], 100.0 * r["int_mismatch_creatures"] / t, ATOL, RTOL,
                 r["float_out_of_tol_creatures"], 100.0 * r["float_out_of_tol_creatures"] / t, r["max_abs_dev"],
                 r["max_abs_dev_in_tol_creatures"], r["errors"]))
        print("         poses / velocities beyond the tolerance: %d (%.5f %%); per field (creature-transitions beyond it, largest "
              "deviation where the integers agree): %s" % (r["pose_velocity_out_of_tol_creatures"],
              100.0 * r["pose_velocity_out_of_tol_creatures"] / t,
              ", ".join("%s %d / %.2g" % (f, c, m) for f, (c, m) in r["by_field"].items())))
