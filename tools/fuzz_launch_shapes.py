#!/usr/bin/env python3
"""Fuzz on the GPU box: every launch shape of the library may change BETWEEN two step calls of a running world -- the
formulation (tile pipeline / fused step kernel), one or two launches for velocity tiles + position blocks, the tile shape
with a fresh tile plan (planned for the morphologies or the library's default plan), the creature order (host-made, device-made
every N steps, identity), issue priority and its thresholds, TOI bodies per wavefront -- and the state arena must not care: after
a random sequence of such changes interleaved with step calls of random lengths the state equals the oracle's run of the same
number of steps in every bit.

    python tools/fuzz_launch_shapes.py [--rounds 40] [--seed 1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def fuzz(rounds, seed, creatures, report=print):
    """Returns the number of rounds whose final state differed from the oracle's (or flagged a solver error)."""
    import torch
    from gym_rem2d_amd import _lib, make_terrain, synthetic
    from gym_rem2d_amd.compiler import Morphology, lanes_for
    from gym_rem2d_amd.world import BatchedWorld
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(seed)
    terrains = [("rough", make_terrain(4)), ("hardcore", make_terrain(4, hardcore=True))]
    bad = 0
    for rnd in range(rounds):
        tname, terrain = terrains[rnd % 2]
        xs, ys, polys = terrain.f32()
        ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
        seed0 = int(rng.integers(0, 10 ** 6))
        specs = synthetic.lsystem_specs(range(seed0, seed0 + creatures), mutate_odd=True)
        lanes = int(rng.choice([4, 8, 16]))
        specs = [s for s in specs if lanes // 2 < s.n_bodies <= lanes] or [s for s in specs if s.n_bodies <= lanes]
        morph = Morphology.from_specs(specs, lanes)
        flags = int(rng.choice([1, 1, 1, 0, 3, 5]))
        w = BatchedWorld(morph.n_envs, morph.lanes, flags)
        w.set_terrain(terrain)
        w.reset(morph, tile_shape=int(rng.choice([0, 1, 2, 3, 4])))
        T, log = 0, []
        for _ in range(int(rng.integers(4, 10))):
            for _ in range(int(rng.integers(0, 4))):   # a few launch-shape changes
                op = int(rng.integers(0, 8))
                if op == 0:
                    v = int(rng.choice([0, 3])); w.set_option("pipeline", v); log.append("pipeline=%d" % v)
                elif op == 1:
                    v = int(rng.integers(0, 3)); w.set_option("fuse_velpost", v); log.append("fuse=%d" % v)   # (2 = the step train)
                elif op == 2:
                    sh = int(rng.integers(0, 5))
                    if rng.integers(0, 2):   # the library's default plan for the shape
                        _lib.check(w.L.rem2d_world_set_tile_shape(w.h, sh))
                        w.tile_shape = sh
                        log.append("shape=%d(default plan)" % sh)
                    else:                    # a plan made for these morphologies
                        _lib.check(w.L.rem2d_world_set_tile_shape(w.h, sh))
                        w.tile_shape = sh
                        tiles = _lib.plan_tiles(morph.arrays["parent"], morph.arrays["jround"], w.n_envs, w.lanes, w.n_envs_padded,
                                                tile_shape=sh)
                        _lib.check(w.L.rem2d_world_set_tiles(w.h, tiles.ctypes.data, len(tiles) - 1))
                        log.append("shape=%d(planned)" % sh)
                elif op == 3:
                    if rng.integers(0, 3) == 0:
                        w.set_order(None); log.append("order=identity")
                    else:
                        w.set_order(torch.from_numpy(rng.permutation(morph.n_envs))); log.append("order=random")
                elif op == 4:
                    v = int(rng.choice([0, 1, 3, 7, 50])); w.set_option("rebalance", v); log.append("rebalance=%d" % v)
                elif op == 5:
                    v = int(rng.integers(0, 6)); w.set_option("prio", v); log.append("prio=%d" % v)
                elif op == 6:
                    a = int(rng.integers(10, 90)); b = int(rng.integers(a, 100))
                    w.set_option("prio_t1", a); w.set_option("prio_t2", b); log.append("prio_t=%d/%d" % (a, b))
                else:
                    v = int(rng.integers(1, 4)); w.set_option("heavy_per_wave", v); log.append("heavy=%d" % v)
            n = int(rng.choice([1, 2, 7, 25, 40, 60]))
            w.step(n)
            T += n
            log.append("step %d" % n)
        got = w.bodies()
        fit = w.view("fitness").cpu().numpy()
        err = int(w.view("err").max())
        w.close()
        ref = O.batch_run(ot, morph.as_dict(), T, n_threads=os.cpu_count() or 1, flags=flags)
        same = err == 0 and np.array_equal(got, ref["bodies"]) and np.array_equal(fit, ref["fitness"])
        bad += 0 if same else 1
        report(json.dumps(dict(round=rnd, terrain=tname, creatures=morph.n_envs, lanes=lanes, flags=flags, steps=T, err=err, equal=bool(same),
                               ops=log if not same else len(log))))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--creatures", type=int, default=600)
    args = ap.parse_args()
    bad = fuzz(args.rounds, args.seed, args.creatures, report=lambda line: print(line, flush=True))
    print("FUZZ " + ("OK" if bad == 0 else "MISMATCH in %d rounds" % bad))
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
