#!/usr/bin/env python3
"""One-off wide parity sweep on the GPU box: thousands of random creatures, HIP path vs the oracle (all host
threads), full state compared with ==.  The unit tests use a few dozen creatures; this looks for rare paths
(TOI events, many pairs per body, deep position iterations, 3-4 joints on a body) in bulk.

    python tools/soak_parity.py [--n 20000] [--steps 300]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--encodings", action="store_true", help="also network-encoded (hardcore track) and direct-encoding populations")
    ap.add_argument("--rebalance", type=int, default=37,
                    help="launch option `rebalance`: the library re-makes every world's creature order on the device every N env-steps "
                         "(0 = off); the default is an odd cadence so that the orders change at many different step numbers")
    ap.add_argument("--flags", type=int, default=1, help="world flags, bits 1 | 2 | 4 (continuous, sleep-reset-always, no-sleep): the same "
                                                         "bits mean the same on both sides")
    ap.add_argument("--wide", action="store_true", help="the wide-slot build librem2d_wide.so (the overflow fallback's library)")
    args = ap.parse_args()
    assert args.flags & ~7 == 0
    import bench
    from gym_rem2d_amd import make_terrain
    from gym_rem2d_amd.compiler import Morphology
    from gym_rem2d_amd.population import LSystemPopulation
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(2026)
    report = []
    cases = [("lsystem/rough", make_terrain(4), 15, 4), ("lsystem40/hardcore", make_terrain(4, hardcore=True), 40, 6),
             ("lsystem/flat", make_terrain(4, flat=True), 20, 0)]
    if args.encodings:
        # the other two encodings: network-encoded creatures (config 4's generator, seeds from 10^6) on the hardcore track and
        # direct-encoding creatures (config 1's) on the default terrain
        cases += [("network/hardcore", make_terrain(4, hardcore=True), -1, 0), ("direct/rough", make_terrain(4), -2, 0)]
    for name, terrain, max_modules, mutate in cases:
        if max_modules == -1:
            from gym_rem2d_amd import synthetic
            batches = [(m, idx.tolist()) for m, idx in synthetic.cppn_batches_native(range(10 ** 6, 10 ** 6 + args.n // 2), n_proc=1)]
        elif max_modules == -2:
            from gym_rem2d_amd import synthetic
            from gym_rem2d_amd.compiler import lanes_for
            specs = synthetic.direct_specs(range(10 ** 6, 10 ** 6 + args.n // 8))
            groups = {}
            for e, sp in enumerate(specs):
                groups.setdefault(lanes_for(sp.n_bodies), []).append(e)
            batches = [(Morphology.from_specs([specs[e] for e in groups[k]], k), groups[k]) for k in sorted(groups)]
        else:
            pop = LSystemPopulation.random(args.n if max_modules < 40 else args.n // 4, rng, max_modules=max_modules)
            for _ in range(mutate):
                pop.mutate(0.3, 0.3, 0.2, rng)
            batches = pop.compile()
        import torch
        from gym_rem2d_amd.world import BatchedWorld
        xs, ys, polys = terrain.f32()
        ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
        bad = 0
        t0 = time.time()
        stats = dict(toi_events=0, max_pairs=0, max_positers=0, creatures=0, bodies=0)
        for m, idx in batches:
            w = BatchedWorld(m.n_envs, m.lanes, args.flags, wide=args.wide,
                             options={"rebalance": args.rebalance} if args.rebalance > 0 else None)
            w.set_terrain(terrain)
            w.reset(m)
            w.step(args.steps)
            got = w.bodies()
            fit = w.view("fitness").cpu().numpy()
            stats["toi_events"] += int(w.view("toievents").sum())
            stats["max_pairs"] = max(stats["max_pairs"], int(w.view("ccount").max()))
            stats["max_positers"] = max(stats["max_positers"], int(w.view("positers").max()))
            assert int(w.view("err").max()) == 0, "solver / pair overflow flagged"
            w.close()
            ref = O.batch_run(ot, m.as_dict(), args.steps, n_threads=os.cpu_count() or 1, flags=args.flags)
            same = np.array_equal(got, ref["bodies"]) and np.array_equal(fit, ref["fitness"])
            if not same:
                bad += int((~(got == ref["bodies"]).all(axis=(1, 2))).sum())
            stats["creatures"] += m.n_envs
            stats["bodies"] += int(m.n_bodies.sum())
        report.append(dict(case=name, steps=args.steps, flags=args.flags, wide=args.wide, rebalance_every=args.rebalance, mismatching_creatures=bad,
                           seconds=round(time.time() - t0, 1), **stats))
        print(json.dumps(report[-1]), flush=True)
    ok = all(r["mismatching_creatures"] == 0 for r in report)
    print("SOAK " + ("OK" if ok else "MISMATCH"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
