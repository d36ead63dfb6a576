"""HIP stepper vs the CPU oracle, through the C ABI, on a real MI355X (pytest -m gpu).

Bar: bit-exact -- integers (pair lists, feature keys, point counts, limit states, awake flags)
AND floats (poses, velocities, impulses): both sides evaluate the same binary32 expression
sequence without FMA contraction, so `==` is the tolerance.  (-0.0 == +0.0 counts as equal.)
"""
import os

import numpy as np
import pytest

from conftest import oracle_terrain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    from gym_rem2d_amd.world import BatchedWorld
    return BatchedWorld


def _populations():
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology
    ls = synthetic.lsystem_specs(range(48), mutate_odd=True)
    return {
        "chain4_top": synthetic.chain_population(20, 4, "top"),
        "chain4_left": synthetic.chain_population(16, 4, "left"),
        "chain4_right": synthetic.chain_population(7, 4, "right"),
        "chain8_top": synthetic.chain_population(9, 8, "top"),
        "lsystem_k32": Morphology.from_specs(ls, 32),
        "lsystem_k16": Morphology.from_specs([s for s in ls if s.n_bodies <= 16], 16),
        "direct": Morphology.from_specs(synthetic.direct_specs(range(40))),
        "pairs_k2": Morphology.from_specs([s for s in synthetic.lsystem_specs(range(60)) if s.n_bodies <= 2], 2),
    }


def _run_gpu(BatchedWorld, morph, terrain, chunks, flags=0):
    w = BatchedWorld(morph.n_envs, morph.lanes, flags)
    w.set_terrain(terrain)
    w.reset(morph)
    snaps = []
    for c in chunks:
        w.step(c)
        snaps.append(w.bodies())
    return w, snaps


@pytest.mark.parametrize("name", ["chain4_top", "chain4_left", "chain4_right", "chain8_top", "lsystem_k32",
                                  "lsystem_k16", "direct", "pairs_k2"])
@pytest.mark.parametrize("terrain_name", ["flat", "rough"])
def test_trajectory_bit_exact(gpu, oracle, flat_terrain, rough_terrain, name, terrain_name):
    morph = _populations()[name]
    terrain = flat_terrain if terrain_name == "flat" else rough_terrain
    chunks = [1, 1, 1, 7, 40, 150, 100]
    T = sum(chunks)
    ref = oracle.batch_run(oracle_terrain(oracle, terrain), morph.as_dict(), T, n_threads=8, trace=True)
    w, snaps = _run_gpu(gpu, morph, terrain, chunks)
    t = 0
    for c, s in zip(chunks, snaps):
        t += c
        assert np.array_equal(s[..., :3], ref["trace"][t - 1]), "pose mismatch at step %d" % t
    assert np.array_equal(snaps[-1], ref["bodies"])  # + velocities, sleep time, awake flag
    assert np.array_equal(w.view("reward").cpu().numpy(), ref["reward"].astype(np.float32))
    assert np.array_equal(w.view("everdone").cpu().numpy(), ref["done"])
    assert np.array_equal(w.view("fitness").cpu().numpy(), ref["fitness"])
    assert int(w.view("err").max()) == 0
    w.close()


def test_contact_and_joint_indices_bit_exact(gpu, oracle, rough_terrain):
    """Pair lists in list order, manifold point counts/types, feature keys, warm-start impulses,
    joint impulses / limit states / motor speeds, island ordering inputs."""
    morph = _populations()["lsystem_k32"]
    ot = oracle_terrain(oracle, rough_terrain)
    for T in (1, 30, 120):
        w, _ = _run_gpu(gpu, morph, rough_terrain, [T])
        cedge, cinfo = w.view("cedge").cpu().numpy(), w.view("cinfo").cpu().numpy()
        key = [w.view("ckey0").cpu().numpy(), w.view("ckey1").cpu().numpy()]
        imp = [w.view(k).cpu().numpy() for k in ("cn0", "cn1", "ct0", "ct1")]
        ccount = w.view("ccount").cpu().numpy()
        jst = [w.view(k).cpu().numpy() for k in ("jimpx", "jimpy", "jimpz", "jmotorimp", "jmotorspeed")]
        jlim = w.view("jlimit").cpu().numpy()
        for e in range(0, morph.n_envs, 3):
            ow = oracle.World.from_morph(ot, morph.as_dict(), e)
            for _ in range(T):
                ow.env_step()
            slots = [s for s in range(morph.lanes) if morph["shape"][e * morph.lanes + s] != 0]
            for b, s in enumerate(slots):
                oc, of = ow.contacts(b)
                assert ccount[e, s] == len(oc)
                for k in range(len(oc)):
                    assert cedge[k, e, s] == oc[k][0]                      # terrain edge, list order
                    n = oc[k][1]
                    assert (cinfo[k, e, s] & 0xff) == n                    # manifold point count
                    if n > 0:
                        assert ((cinfo[k, e, s] >> 8) & 0xff) == oc[k][2]  # manifold type
                    for j in range(n):
                        assert (int(key[j][k, e, s]) & 0xffffffff) == (int(oc[k][4 + j]) & 0xffffffff)
                        assert imp[j][k, e, s] == of[k][j] and imp[2 + j][k, e, s] == of[k][2 + j]
            oj = ow.joints()
            for b, s in enumerate(slots[1:]):
                for q in range(5):
                    assert jst[q][e, s] == oj[b][q]
                assert jlim[e, s] == int(oj[b][5])
        w.close()


@pytest.mark.parametrize("name", ["chain4_left", "chain8_top", "lsystem_k32", "direct", "pairs_k2"])
def test_continuous_physics_bit_exact(gpu, oracle, rough_terrain, name):
    """b2World::SolveTOI (continuousPhysics, the pybox2d default): TOI search per pair (b2TimeOfImpact
    / b2Distance), TOI sub-steps with their own position + 180-iteration velocity solve."""
    from gym_rem2d_amd import _lib
    morph = _populations()[name]
    chunks = [1, 2, 7, 40, 150]
    T = sum(chunks)
    ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), morph.as_dict(), T, n_threads=8, trace=True,
                           flags=oracle.FLAG_CONTINUOUS)
    w, snaps = _run_gpu(gpu, morph, rough_terrain, chunks, _lib.FLAG_CONTINUOUS)
    t = 0
    for c, s in zip(chunks, snaps):
        t += c
        assert np.array_equal(s[..., :3], ref["trace"][t - 1]), "pose mismatch at step %d" % t
    assert np.array_equal(snaps[-1], ref["bodies"])
    assert np.array_equal(w.view("fitness").cpu().numpy(), ref["fitness"])
    assert int(w.view("err").max()) == 0
    assert int(w.view("toievents").sum()) > 0  # landings do trigger TOI sub-steps
    disc = oracle.batch_run(oracle_terrain(oracle, rough_terrain), morph.as_dict(), T, n_threads=8)
    assert not np.array_equal(disc["bodies"], ref["bodies"])  # and they change the outcome
    w.close()


@pytest.mark.parametrize("continuous", [False, True])
def test_hardcore_terrain_bit_exact(gpu, oracle, continuous):
    """bipedal-walker-hardcore track (pits / stumps / stairs = static boxes): b2CollidePolygons and
    b2CollidePolygonAndCircle against terrain boxes, box proxies ahead of the edges in pair order.
    Creatures are dropped along the track so that they actually land on the obstacles."""
    from gym_rem2d_amd import _lib, make_terrain, synthetic
    from gym_rem2d_amd.compiler import Morphology
    terrain = make_terrain(4, hardcore=True)
    assert len(terrain.polys) == 29
    specs = synthetic.lsystem_specs(range(40), mutate_odd=True)
    morph = Morphology.from_specs(specs, 32)
    # spread the creatures over the first obstacles: shift x by a multiple of 1.4 m (positions are float32 inputs)
    xs = terrain.polys[:, :, 0]
    first = float(xs.min())
    shift = (np.arange(morph.n_envs) % 20) * 1.4 + (first - 7.0)
    a = morph.arrays
    K = morph.lanes
    for e in range(morph.n_envs):
        sl = slice(e * K, (e + 1) * K)
        live = a["shape"][sl] != 0
        a["x"][sl][live] = (a["x"][sl][live] + np.float32(shift[e])).astype(np.float32)
        a["y"][sl][live] = (a["y"][sl][live] + np.float32(2.0)).astype(np.float32)
    flags_g = _lib.FLAG_CONTINUOUS if continuous else 0
    flags_o = oracle.FLAG_CONTINUOUS if continuous else 0
    T = 260
    ref = oracle.batch_run(oracle_terrain(oracle, terrain), morph.as_dict(), T, n_threads=8, trace=True, flags=flags_o)
    w, snaps = _run_gpu(gpu, morph, terrain, [1, 9, 50, 200], flags_g)
    t = 0
    for c, sn in zip([1, 9, 50, 200], snaps):
        t += c
        assert np.array_equal(sn[..., :3], ref["trace"][t - 1]), "pose mismatch at step %d" % t
    assert np.array_equal(snaps[-1], ref["bodies"])
    assert int(w.view("err").max()) == 0
    # the boxes are really hit: some pair slots reference static proxies below the edge range
    cedge, cinfo = w.view("cedge").cpu().numpy(), w.view("cinfo").cpu().numpy()
    hit = (cedge >= 0) & (cedge < 29) & ((cinfo & 0xff) > 0)
    assert hit.sum() > 0
    w.close()


def test_reference_default_size_creatures_k64(gpu, oracle, rough_terrain):
    """0.cfg allows max_size = 40 (up to 41 bodies): one creature per wavefront (64 lanes)."""
    import random
    from gym_rem2d_amd import _lib, build_creature, ea
    from gym_rem2d_amd.compiler import Morphology
    random.seed(1)
    cfg = ea.make_config(population_size=64, encoding="lsystem")
    specs = []
    for _ in range(64):
        ind = ea.Individual.random(config=cfg)
        spec = build_creature(ind.genome.create(7).getNodes(), ind.genome.moduleList)[0]
        if spec.n_bodies > 32:
            specs.append(spec)
    assert len(specs) >= 5 and max(s.n_bodies for s in specs) >= 40
    morph = Morphology.from_specs(specs)
    assert morph.lanes == 64
    ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), morph.as_dict(), 150, n_threads=8,
                           flags=oracle.FLAG_CONTINUOUS)
    w, snaps = _run_gpu(gpu, morph, rough_terrain, [1, 49, 100], _lib.FLAG_CONTINUOUS)
    assert np.array_equal(snaps[-1], ref["bodies"])
    assert int(w.view("err").max()) == 0
    w.close()


def test_multi_step_launch_equals_single_steps(gpu, rough_terrain):
    morph = _populations()["direct"]
    _, a = _run_gpu(gpu, morph, rough_terrain, [1] * 60)
    _, b = _run_gpu(gpu, morph, rough_terrain, [60])
    _, c = _run_gpu(gpu, morph, rough_terrain, [13, 47])
    assert np.array_equal(a[-1], b[-1]) and np.array_equal(a[-1], c[-1])


def test_sleep_variants_and_single_bodies(gpu, oracle, flat_terrain):
    """Single-module creatures fall, settle and go to sleep for good (no joint to wake them);
    both SetAwake variants and doSleep=False follow the oracle."""
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.compiler import Morphology
    singles = [s for s in synthetic.lsystem_specs(range(80)) if s.n_bodies == 1][:8]
    multi = synthetic.lsystem_specs(range(5))
    morph = Morphology.from_specs(singles + multi, 32)
    for flags in (0, _lib.FLAG_SLEEP_RESET_ALWAYS, _lib.FLAG_NO_SLEEP):
        ref = oracle.batch_run(oracle_terrain(oracle, flat_terrain), morph.as_dict(), 200, n_threads=8, flags=flags)
        w, snaps = _run_gpu(gpu, morph, flat_terrain, [200], flags)
        assert np.array_equal(snaps[-1], ref["bodies"])
        if flags == 0:
            assert np.all(snaps[-1][:len(singles), 0, 7] == 0)  # asleep
        w.close()


def test_padding_and_empty_lanes(gpu, oracle, rough_terrain):
    from gym_rem2d_amd import synthetic
    morph = synthetic.chain_population(5, 3, "top", lanes=4)  # 5 envs -> padded to 16, lane 3 empty
    ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), morph.as_dict(), 50, n_threads=2)
    w, snaps = _run_gpu(gpu, morph, rough_terrain, [50])
    assert w.n_envs_padded == 16 and snaps[-1].shape == (5, 4, 8)
    assert np.array_equal(snaps[-1], ref["bodies"])
    assert np.all(snaps[-1][:, 3, :] == 0)
    w.close()


def test_full_size_replicas_agree(gpu, oracle, flat_terrain):
    """BASELINE full size (65 536 creatures): size-independent properties.  Identical creatures
    stay identical in every env (checksum of checksums), and equal the oracle's single env;
    a tiled heterogeneous population reproduces its 256 unique members in every tile."""
    import torch
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology
    N = 65536
    morph = synthetic.chain_population(N, 4, "left")
    w, _ = _run_gpu(gpu, morph, flat_terrain, [100])
    one = oracle.batch_run(oracle_terrain(oracle, flat_terrain), synthetic.chain_population(1, 4, "left").as_dict(), 100)
    for k in ("px", "py", "ang", "vx", "vy", "w"):
        v = w.view(k)
        assert bool((v == v[0:1]).all())
    assert np.array_equal(w.bodies()[0], one["bodies"][0])
    w.close()
    uniq = Morphology.from_specs([s for s in synthetic.lsystem_specs(range(400)) if s.n_bodies <= 16][:256], 16)
    tiled = uniq.take(np.arange(N) % 256)
    from gym_rem2d_amd import _lib
    w, _ = _run_gpu(gpu, tiled, flat_terrain, [60], _lib.FLAG_CONTINUOUS)
    ref = oracle.batch_run(oracle_terrain(oracle, flat_terrain), uniq.as_dict(), 60, n_threads=8,
                           flags=oracle.FLAG_CONTINUOUS)
    px = w.view("px").view(N // 256, 256, 16)
    assert bool((px == px[0:1]).all())
    assert np.array_equal(w.bodies()[:256], ref["bodies"])
    assert int(w.view("err").max()) == 0
    w.close()


def test_full_size_chain8(gpu, oracle, flat_terrain):
    """north_star's target workload at full size: 65 536 x 8-module chains through BatchedModular2D -- the automatic
    128-lane tiles and three step groups (one rem2d_groups_step call per step call) that bench.py's `chain8` workload
    measures.  Creature 0 equals the oracle in every bit, every replica equals creature 0, no error bits."""
    import torch
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.env import BatchedModular2D
    N, steps = 65536, 200
    env = BatchedModular2D(flat=True, flags=_lib.FLAG_CONTINUOUS)
    env.reset_morphology(synthetic.chain_population(N, 8, "left"))
    import os
    assert len(env.groups) == 3 and len(env.worlds) == 3
    assert env._tile_shape_used == 4 or "REM2D_TILE_SHAPE" in os.environ   # (the suite is also run with the shape forced)
    for _ in range(steps // 25):
        env.step(25)
    torch.cuda.synchronize()
    assert int(env.errors().max()) == 0
    one = oracle.batch_run(oracle_terrain(oracle, flat_terrain), synthetic.chain_population(1, 8, "left").as_dict(), steps,
                           flags=oracle.FLAG_CONTINUOUS)
    seen = 0
    for w, idx in env.worlds:
        b = w.bodies()
        assert np.array_equal(b, np.broadcast_to(one["bodies"][0], b.shape))       # creature 0 == oracle, replicas == creature 0
        seen += b.shape[0]
    assert seen == N
    fit = env.fitness.cpu().numpy()
    assert np.array_equal(fit, np.full(N, one["fitness"][0]))
    assert bool((env.steps == steps).all())
    env.close()


@pytest.mark.parametrize("launch", ["velpost", "two_launches", "tiles_128_bodies", "tiles_128_static"])
@pytest.mark.parametrize("skip_frozen", [False, True])
def test_post_kernel_retiling_keeps_every_bit(gpu, oracle, rough_terrain, skip_frozen, launch):
    """REM2D_FLAG_RETILE: the creatures are dealt to the wavefronts anew in every step (those that used all 60 position
    iterations first; the order comes out of atomics and differs from run to run) -- since round 4 for the velocity tiles as
    for the position blocks (tile slot e = creature order[e]), in the one-launch form and in the two-launch forms alike (a
    static tile shape keeps the arena order in its velocity tiles: the host planned them for it).
    Creatures are independent:
    poses, velocities, sleep state, reward, fitness and step counts equal the oracle's in every bit, also together with
    REM2D_FLAG_SKIP_FROZEN (wavefronts of pre whose creatures have all finished are skipped; post must leave exactly those
    creatures alone, whichever of its wavefronts they ride in)."""
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.compiler import Morphology
    specs = [s for s in synthetic.lsystem_specs(range(300), mutate_odd=True) if s.n_bodies <= 8]
    morph = Morphology.from_specs(specs, 8)
    flags = _lib.FLAG_CONTINUOUS | _lib.FLAG_RETILE | (_lib.FLAG_SKIP_FROZEN if skip_frozen else 0)
    T = 420 if skip_frozen else 240     # (long enough for whole wavefronts of 8 creatures to have finished)
    w = gpu(morph.n_envs, morph.lanes, flags, options={"fuse_velpost": 0} if launch == "two_launches" else None)
    w.set_terrain(rough_terrain)
    w.reset(morph, tile_shape={"tiles_128_bodies": 1, "tiles_128_static": 4}.get(launch))
    snaps = []
    for c in [1, 59, T - 60]:
        w.step(c)
        snaps.append(w.bodies())
    ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), morph.as_dict(), T, n_threads=8, flags=oracle.FLAG_CONTINUOUS)
    assert np.array_equal(w.view("fitness").cpu().numpy(), ref["fitness"])
    assert int(w.view("err").max()) == 0
    order = w.view("positers").cpu().numpy()
    assert (order == 60).any() and (order < 60).any()          # both classes of creatures occur
    if not skip_frozen:
        assert np.array_equal(snaps[-1], ref["bodies"])
        assert np.array_equal(w.view("reward").cpu().numpy(), ref["reward"].astype(np.float32))
        assert bool((w.view("steps") == T).all())
    else:
        # creatures of skipped wavefronts stop where they were; the others are the oracle's
        frozen = w.view("frozen").cpu().numpy() != 0
        moved = w.view("steps").cpu().numpy() == T
        assert frozen.any() and np.array_equal(snaps[-1][moved], ref["bodies"][moved]) and (moved | frozen).all()
    w.close()


@pytest.mark.parametrize("launch", ["step_train", "velpost", "two_launches", "tiles_128_bodies", "tiles_128_static", "tiles_256_static",
                                    "fused_step_kernel"])
def test_host_creature_order_keeps_every_bit(gpu, oracle, rough_terrain, launch):
    """rem2d_world_set_order: slot e of the velocity tiles / position blocks handles creature order[e].  Any permutation --
    here a random one, changed twice in mid-run, then back to the identity -- leaves poses, velocities, sleep state,
    reward and fitness equal to the oracle's in every bit (creatures are independent), in the one-launch form, the two
    launches, the 128-body tiles, the static tile shapes (whose velocity tiles were planned for the arena order and keep it:
    only their position blocks follow the order) and (where the order has no meaning and is ignored) the fused step kernel."""
    import torch
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.compiler import Morphology
    specs = [s for s in synthetic.lsystem_specs(range(400), mutate_odd=True) if 3 <= s.n_bodies <= 8]
    morph = Morphology.from_specs(specs, 8)
    opts = {"two_launches": {"fuse_velpost": 0}, "velpost": {"fuse_velpost": 1}, "fused_step_kernel": {"pipeline": 0}}.get(launch)
    w = gpu(morph.n_envs, morph.lanes, _lib.FLAG_CONTINUOUS, options=opts)
    w.set_terrain(rough_terrain)
    shape = {"tiles_128_bodies": 1, "tiles_128_static": 4, "tiles_256_static": 0}.get(launch)
    w.reset(morph, tile_shape=shape)
    g = torch.Generator().manual_seed(7)
    T = 0
    for chunk, order in ((40, torch.randperm(morph.n_envs, generator=g)), (80, torch.randperm(morph.n_envs, generator=g)),
                         (60, torch.arange(morph.n_envs).flip(0)), (40, None)):
        w.set_order(order)
        w.step(chunk)
        T += chunk
    ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), morph.as_dict(), T, n_threads=8, flags=oracle.FLAG_CONTINUOUS)
    assert np.array_equal(w.bodies(), ref["bodies"])
    assert np.array_equal(w.view("fitness").cpu().numpy(), ref["fitness"])
    assert np.array_equal(w.view("reward").cpu().numpy(), ref["reward"].astype(np.float32))
    assert int(w.view("err").max()) == 0 and int(w.view("toievents").sum()) > 0
    with pytest.raises(ValueError, match="permutation"):        # the host wrapper refuses what is not a permutation
        w.set_order(torch.zeros(morph.n_envs, dtype=torch.long))
    with pytest.raises(ValueError, match="permutation"):
        w.set_order(torch.arange(morph.n_envs - 1))
    w.close()
    r = gpu(morph.n_envs, morph.lanes, _lib.FLAG_CONTINUOUS | _lib.FLAG_RETILE)
    with pytest.raises(_lib.Rem2dError, match="RETILE"):     # a world that deals its creatures itself takes no host order
        r.set_order(torch.arange(morph.n_envs))
    with pytest.raises(_lib.Rem2dError, match="RETILE"):
        r.set_option("rebalance", 10)
    r.close()
    # the same order made by the library itself every few steps (REM2D_OPT_REBALANCE: a stable partition on the device, the
    # creatures that used every position iteration first)
    opts = dict(opts or {}, rebalance=7)
    d = gpu(morph.n_envs, morph.lanes, _lib.FLAG_CONTINUOUS, options=opts)
    d.set_terrain(rough_terrain)
    d.reset(morph, tile_shape=shape)
    for chunk in (40, 80, 60, 40):
        d.step(chunk)
    assert np.array_equal(d.bodies(), ref["bodies"]) and np.array_equal(d.view("fitness").cpu().numpy(), ref["fitness"])
    assert int(d.view("err").max()) == 0
    d.close()


def test_launch_shapes_may_change_between_step_calls(gpu, oracle):
    """The state arena does not care how it is stepped: formulation, one or two launches, tile shape and tile plan, creature
    order (host- / device-made), issue priority and TOI bodies per wavefront are changed at random between step calls of
    random lengths (tools/fuzz_launch_shapes.py, a few rounds of it; 340 rounds: profiles/r04_fuzz_launch_shapes.txt) and the
    final state still equals the oracle's run of the same number of steps in every bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_launch_shapes
    lines = []
    assert fuzz_launch_shapes.fuzz(8, 5, 500, report=lines.append) == 0, lines


def test_determinism(gpu, rough_terrain):
    morph = _populations()["lsystem_k16"]
    _, a = _run_gpu(gpu, morph, rough_terrain, [150])
    _, b = _run_gpu(gpu, morph, rough_terrain, [150])
    assert np.array_equal(a[-1], b[-1])


def test_call_order_errors(gpu):
    from gym_rem2d_amd import _lib, synthetic
    w = gpu(4, 4)
    with pytest.raises(_lib.Rem2dError, match="set_terrain"):
        w.step(1)
    with pytest.raises(ValueError):
        w.reset(synthetic.chain_population(3, 4))
    w.close()


def test_gpu_matches_committed_trajectory_digests(gpu):
    """Golden vectors without the oracle in the loop: the HIP path must reproduce the committed SHA-256
    digests of poses, velocities, sleep timers, reward, done and fitness (tests/golden/trajectory_digest.json)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_trajectory_digest as D
    with open(os.path.join(root, "tests", "golden", "trajectory_digest.json")) as f:
        gold = json.load(f)["cases"]
    for name, pop, ter, flags, steps in D.CASES:
        m, t = D.population(pop), D.terrain(ter)
        w = gpu(m.n_envs, m.lanes, flags)
        w.set_terrain(t)
        w.reset(m)
        w.step(steps)
        got = D.digest(w.bodies(), m.n_bodies, w.view("reward").cpu().numpy(), w.view("everdone").cpu().numpy(),
                       w.view("fitness").cpu().numpy())
        assert int(w.view("err").max()) == 0
        w.close()
        assert got == gold[name]["sha256"], name


@pytest.mark.parametrize("vel_iters,pos_iters,flags", [(8, 3, 0), (40, 100, 1), (3, 70, 0)])
def test_step_ex_iteration_counts_bit_exact(gpu, oracle, rough_terrain, vel_iters, pos_iters, flags):
    """rem2d_world_step_ex with Box2D's usual (8, 3) and with more than 64 position iterations (the pipelined
    position solver keeps its per-iteration flags in a 64-entry ring) against the oracle's env_step_ex."""
    morph = _populations()["lsystem_k16"]
    ot = oracle_terrain(oracle, rough_terrain)
    steps = 120
    d = morph.as_dict()
    refs = []
    for e in range(morph.n_envs):
        ow = oracle.World.from_morph(ot, d, e, flags)
        for _ in range(steps):
            ow.env_step_ex(1.0 / 50, vel_iters, pos_iters)
        refs.append((ow.bodies(), ow.position_iterations))
    # the default 64-body tiles, the 128-body flexible and static shapes, velocity tiles + position blocks in two launches
    for shape, opts in ((None, None), (1, None), (4, None), (None, {"fuse_velpost": 0}), (None, {"fuse_velpost": 1}), (None, {"rebalance": 9}),
                        (None, {"rebalance": 9, "fuse_velpost": 1})):
        w = gpu(morph.n_envs, morph.lanes, flags, options=opts)
        w.set_terrain(rough_terrain)
        w.reset(morph, tile_shape=shape)
        w.step_ex(steps, 1.0 / 50, vel_iters, pos_iters)
        got = w.bodies()
        used = w.view("positers").cpu().numpy()
        assert int(w.view("err").max()) == 0
        w.close()
        for e, (ref, iters) in enumerate(refs):
            assert np.array_equal(got[e, : ref.shape[0]], ref), (e, shape, opts)
            assert used[e] == iters
    print('position iterations used: max %d of %d' % (used.max(), pos_iters))
    assert used.max() == pos_iters or pos_iters > 60   # with the usual budgets some creature runs out of iterations


@pytest.mark.parametrize("variant", [{"pipeline": 0}, {"tile_shape": 0}, {"tile_shape": 1}, {"tile_shape": 2}, {"tile_shape": 4}, {"fuse_velpost": 0},
                                     {"fuse_velpost": 1}, {"prio": 0, "heavy_per_wave": 2, "fuse_velpost": 1}, {"prio": 0}],
                         ids=["fused_step_kernel", "tiles_256_bodies", "tiles_128_bodies", "tiles_192_bodies", "tiles_128_bodies_static_sets",
                              "velocity_and_position_in_two_launches", "one_launch_per_step_velpost",
                              "no_issue_priority_two_toi_bodies_per_wavefront", "step_train_without_issue_priority"])
def test_other_formulations_match_committed_digests(gpu, variant):
    """The library's switches are per-world launch options (rem2d_world_set_option / rem2d_world_set_tile_shape; it reads no
    environment variable), so every other formulation runs in this very process: the fused body-per-lane kernel of round 1
    (pipeline 0) and the wider tile shapes of the velocity kernel (256 / 192 / 128 bodies per wavefront, 4 / 3 / 2 joint
    register sets; the latter two with flexible joint placement, rotation and period choice like the default) reproduce the same committed digests as the default (which runs a block's velocity tiles and its position
    iterations in one launch, rem2d_velpost_kernel); so do the two launches rem2d_vel4_kernel + rem2d_post_multi_kernel, and
    so does the default formulation without its scheduling hints (issue priority, one TOI body per wavefront)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_trajectory_digest as D
    variant = dict(variant)
    shape = variant.pop("tile_shape", None)
    got = {}
    for name, pop, ter, flags, steps in D.CASES:
        m, t = D.population(pop), D.terrain(ter)
        w = gpu(m.n_envs, m.lanes, flags, options=variant)
        assert all(w.get_option(k) == v for k, v in variant.items())
        w.set_terrain(t)
        w.reset(m, tile_shape=shape)
        w.step(steps)
        got[name] = D.digest(w.bodies(), m.n_bodies, w.view('reward').cpu().numpy(), w.view('everdone').cpu().numpy(),
                             w.view('fitness').cpu().numpy())
        assert int(w.view("err").max()) == 0
        w.close()
    with open(os.path.join(root, "tests", "golden", "trajectory_digest.json")) as f:
        gold = json.load(f)["cases"]
    assert got == {k: v["sha256"] for k, v in gold.items()}


def test_option_argument_errors(gpu):
    import ctypes as C
    from gym_rem2d_amd import _lib
    w = gpu(4, 4, 0)
    L = _lib.lib()
    assert [w.get_option(k) for k in _lib.OPTIONS] == [3, 2, 5, 60, 75, 1, 0, 0, 0]     # the documented defaults
    for key, bad in ((0, 1), (0, 2), (1, 3), (5, 0), (5, 65), (2, -1), (99, 0), (-1, 0)):
        assert L.rem2d_world_set_option(w.h, key, bad) == -1, (key, bad)
    assert b"option" in L.rem2d_last_error()
    assert L.rem2d_world_get_option(w.h, 99, C.byref(C.c_int32())) == -1
    w.set_option("heavy_per_wave", 64)
    assert w.get_option("heavy_per_wave") == 64
    w.close()


def test_scalar_helpers_special_values(gpu, oracle):
    """b2Min / b2Max / b2Clamp are one v_med3_f32 each on the GPU (rem2d_math.h) and ``a < b ? a : b`` chains in the oracle
    (Box2D's own form).  For every pair / triple of ordinary numbers the two give the same bits; this test feeds both sides
    the values no trajectory reaches -- NaN, +-inf, denormals, +-0, +-FLT_MAX -- through rem2d_selftest_scalar and
    rem2d_oracle_kat_scalar and pins down WHERE they may differ, and nowhere else:
      (1) a result that is a zero may carry the other sign (the median keeps the argument's -0 where Box2D's form hands
          out a bound's +0; -0 == +0 in every comparison downstream and the trajectory digests canonicalise it);
      (2) an operand that is NaN (Box2D's form returns whichever operand the failed comparison selects, the median
          returns a number) -- a state that holds a NaN has left the specification anyway, and every parity test compares
          with array_equal, which fails on NaN.
    Denormals are kept on both sides (no flush to zero), infinities clamp like numbers.  The same call checks b2Rot::Set
    ("rem2d trig") of the device against the oracle's on 200 000 angles incl. huge and denormal ones, bit for bit."""
    import ctypes as C
    import torch
    from gym_rem2d_amd import _lib
    tiny = np.float32(1e-45)
    S = np.array([0.0, -0.0, tiny, -tiny, 1e-40, -1e-40, np.finfo(np.float32).tiny, -np.finfo(np.float32).tiny, 1.0, -1.0,
                  0.5, -2.5, 3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan], dtype=np.float32)
    rng = np.random.default_rng(5)
    a, b, c = [v.ravel() for v in np.meshgrid(S, S, S, indexing="ij")]
    ok3 = ~(np.isnan(b) | np.isnan(c)) & (b <= c)                # clamp is only ever called with lo <= hi
    a, b, c = a[ok3], b[ok3], c[ok3]
    ra = (rng.standard_normal(20000) * 10 ** rng.uniform(-6, 6, 20000)).astype(np.float32)
    rb = (rng.standard_normal(20000) * 10 ** rng.uniform(-6, 6, 20000)).astype(np.float32)
    rc = np.maximum(rb, (rng.standard_normal(20000) * 10 ** rng.uniform(-6, 6, 20000)).astype(np.float32))
    a, b, c = np.concatenate([a, ra]), np.concatenate([b, rb]), np.concatenate([c, rc])
    n = len(a)
    dev = torch.device("cuda", 0)
    ta, tb, tc = (torch.from_numpy(v).to(dev) for v in (a, b, c))
    out = torch.zeros(5 * n, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().rem2d_selftest_scalar(ta.data_ptr(), tb.data_ptr(), tc.data_ptr(), n, out.data_ptr(), 0,
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(5, n)
    ref = oracle.kat_scalar(a, b, c)
    bits = lambda x: x.view(np.uint32)
    nan_in = [np.isnan(a) | np.isnan(b), np.isnan(a) | np.isnan(b), np.isnan(a)]
    report = {}
    for row, name in enumerate(("min", "max", "clamp")):
        g, r = got[row], ref[row]
        same = bits(g) == bits(r)
        zero_sign = ~same & (g == 0) & (r == 0)                              # (1)
        nan_case = ~same & nan_in[row]                                       # (2)
        assert (same | zero_sign | nan_case).all(), (name, a[~(same | zero_sign | nan_case)][:5], b[~(same | zero_sign | nan_case)][:5])
        assert not np.isnan(g[~nan_in[row]]).any()
        report[name] = (int(zero_sign.sum()), int(nan_case.sum()))
        # ordinary numbers, denormals and infinities: identical bits, always
        plain = ~nan_in[row] & ~((g == 0) & (r == 0))
        assert (bits(g)[plain] == bits(r)[plain]).all()
    assert got[0][(a == tiny) & (b == 1.0)][0] == tiny and got[2][(a == -tiny) & (b == -1.0) & (c == 1.0)][0] == -tiny   # no flush
    print("accepted differences (zero sign, NaN operand):", report)
    # b2Rot::Set on the device vs the oracle: bit for bit, incl. huge / denormal angles (NaN / inf excluded: rint of them)
    ang = np.concatenate([rng.uniform(-1e3, 1e3, 150000), rng.uniform(-8, 8, 49000), [0.0, -0.0, 1e-45, -1e-40, 1e6, -3e7] + [0.0] * 994]
                         ).astype(np.float32)
    m = len(ang)
    tang = torch.from_numpy(ang).to(dev)
    out2 = torch.zeros(5 * m, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().rem2d_selftest_scalar(tang.data_ptr(), tang.data_ptr(), tang.data_ptr(), m, out2.data_ptr(), 0,
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize()
    g2 = out2.cpu().numpy().reshape(5, m)
    sc = np.array([oracle.sincosf(float(x)) for x in ang[::7]], dtype=np.float32)
    assert np.array_equal(bits(g2[3][::7].copy()), bits(sc[:, 0].copy())) and np.array_equal(bits(g2[4][::7].copy()), bits(sc[:, 1].copy()))


def test_worlds_step_argument_errors(gpu, flat_terrain):
    import ctypes as C
    from gym_rem2d_amd import _lib, synthetic
    m = synthetic.chain_population(4, 4, "top")
    a, b = gpu(4, m.lanes, 0), gpu(4, m.lanes, _lib.FLAG_CONTINUOUS)
    for w in (a, b):
        w.set_terrain(flat_terrain)
        w.reset(m)
    L = _lib.lib()
    st = a._stream()
    two = (C.c_void_p * 2)(a.h, b.h)
    assert L.rem2d_worlds_step(two, 2, 1, st) != 0 and b"CONTINUOUS" in L.rem2d_last_error()
    nine = (C.c_void_p * 9)(*([a.h] * 9))
    assert L.rem2d_worlds_step(nine, 9, 1, st) != 0 and b"too many" in L.rem2d_last_error()
    assert L.rem2d_worlds_step(two, 0, 1, st) != 0
    c = gpu(4, m.lanes, 0)
    pair = (C.c_void_p * 2)(a.h, c.h)
    assert L.rem2d_worlds_step(pair, 2, 1, st) != 0 and b"set_terrain" in L.rem2d_last_error()   # c has no terrain yet
    for w in (a, b, c):
        w.close()


def test_full_size_config4_cppn_on_hardcore_terrain(gpu, oracle):
    """BASELINE config 4 at full size: 65 536 network-encoded creatures on the hardcore track (pits, stumps, stairs:
    polygon-polygon / polygon-circle manifolds and TOI against static boxes), continuous physics, through the env facade
    (lane buckets, tiles, step groups).  384 unique creatures tiled: the first tile equals the oracle in every bit,
    every replica equals the first tile, no solver / pair overflow anywhere."""
    import torch
    from gym_rem2d_amd import _lib, make_terrain, synthetic
    from gym_rem2d_amd.compiler import Morphology, lanes_for
    from gym_rem2d_amd.env import BatchedModular2D
    N, U, steps = 65536, 384, 150
    specs = synthetic.cppn_specs(range(U))
    env = BatchedModular2D(hardcore=True, flags=_lib.FLAG_CONTINUOUS)
    env.reset_specs([specs[e % U] for e in range(N)])
    for _ in range(steps // 50):
        env.step(50)
    torch.cuda.synchronize()
    assert int(env.errors().max()) == 0
    terrain = make_terrain(4, hardcore=True)
    ot = oracle_terrain(oracle, terrain)
    groups = {}
    for e, s in enumerate(specs):
        groups.setdefault(lanes_for(s.n_bodies), []).append(e)
    fit = env.fitness.cpu().numpy()
    ref_fit = np.zeros(U)
    for k in sorted(groups):
        m = Morphology.from_specs([specs[e] for e in groups[k]], k)
        r = oracle.batch_run(ot, m.as_dict(), steps, n_threads=8, flags=oracle.FLAG_CONTINUOUS)
        ref_fit[groups[k]] = r["fitness"]
    assert np.array_equal(fit[:U], ref_fit)                         # evaluate()'s running fitness, float64, first tile
    R = N // U
    assert np.array_equal(fit[:R * U].reshape(R, U), np.broadcast_to(ref_fit, (R, U)))   # every full replica
    steps_taken = env.steps.cpu().numpy()
    assert (steps_taken == steps).all()
    env.close()


def test_full_size_config5_generation_share(gpu, oracle, rough_terrain):
    """BASELINE config 5's per-GPU share: a 131 072-individual array population through the native L-system compiler,
    lane buckets, tiles and whole episodes (evaluate()'s rule, REM2D_FLAG_SKIP_FROZEN).  A 1 % sample of the
    individuals -- plus every individual that needed the wide-slot fallback -- is re-evaluated by the oracle: identical
    float64 fitness for ALL of them."""
    import torch
    from gym_rem2d_amd import _lib
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import check_errors, run_episode
    from gym_rem2d_amd.population import LSystemPopulation
    N, cap = 131072, 300
    rng = np.random.default_rng(5)
    pop = LSystemPopulation.random(N, rng, max_modules=15)
    env = BatchedModular2D(flags=_lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN)
    env._upload(pop.compile(0), N)
    fit = run_episode(env, max_steps=cap).cpu().numpy()      # overflowing creatures are re-evaluated in the wide build
    n_fallback = len(env.last_overflow)
    env.close()
    assert fit.shape == (N,) and np.isfinite(fit).all() and (fit > 0).mean() > 0.5
    sample = rng.choice(N, N // 100, replace=False)
    ot = oracle_terrain(oracle, rough_terrain)
    ref = np.zeros(N)
    fell_back = np.asarray(sorted(set(env.last_overflow) - set(sample.tolist())), dtype=np.int64)
    sample = np.concatenate([sample, fell_back])
    for m, idx in pop.select(sample).compile(0):
        r = oracle.batch_run(ot, m.as_dict(), cap, n_threads=8, flags=oracle.FLAG_CONTINUOUS)
        ref[sample[np.asarray(idx)]] = r["fitness"]
    assert n_fallback < N // 100                                     # overflow of the default slots is rare
    assert np.array_equal(fit[sample], ref[sample])                  # every sampled individual, fallback or not


def test_tile_shape_per_world_and_mixed_in_one_launch(gpu, oracle, rough_terrain):
    """rem2d_world_set_tile_shape: the launch shape of the velocity kernel is a property of the world (the env picks
    the 128-lane tiles for populations beyond ~130 000 creatures).  The five shapes give the oracle's bits, also when
    worlds planned for different shapes share one merged launch (rem2d_worlds_step)."""
    import ctypes as C
    from gym_rem2d_amd import _lib
    pops = _populations()
    morphs = [pops["lsystem_k32"], pops["chain8_top"], pops["direct"]]
    T = 90
    refs = [oracle.batch_run(oracle_terrain(oracle, rough_terrain), m.as_dict(), T, n_threads=8, flags=oracle.FLAG_CONTINUOUS)
            for m in morphs]
    for shapes in ((0, 0, 0), (1, 1, 1), (3, 3, 3), (0, 3, 1), (3, 1, 0), (2, 2, 2), (4, 4, 4), (4, 1, 3), (2, 4, 1)):
        ws = []
        for m, sh in zip(morphs, shapes):
            if m.lanes > (256, 128, 192, 64, 128)[sh]:
                sh = 0
            w = gpu(m.n_envs, m.lanes, flags=_lib.FLAG_CONTINUOUS)
            w.set_terrain(rough_terrain)
            w.reset(m, tile_shape=sh)
            assert w.tile_shape == sh
            ws.append(w)
        arr = (C.c_void_p * len(ws))(*[w.h for w in ws])
        for n in (1, 29, 60):
            _lib.check(_lib.lib().rem2d_worlds_step(arr, len(ws), n, ws[0]._stream()))
        for w, ref in zip(ws, refs):
            assert np.array_equal(w.bodies(), ref["bodies"]), shapes
            assert np.array_equal(w.view("fitness").cpu().numpy(), ref["fitness"])
            assert int(w.view("err").max()) == 0
            w.close()
    # the one unsound mix: a tile planned for shape 2 (<= 192 joints IN ALL, placed flexibly) may hold more than 64 joints in one
    # schedule phase, which the static four-set kernel that a shape-0 world pulls the launch to cannot take -- refused, in every
    # entry point that merges worlds, instead of dropping joints (ABI v9)
    ws = []
    for m, sh in zip(morphs, (2, 3, 0)):
        w = gpu(m.n_envs, m.lanes, flags=_lib.FLAG_CONTINUOUS)
        w.set_terrain(rough_terrain)
        w.reset(m, tile_shape=0 if m.lanes > (256, 128, 192, 64, 128)[sh] else sh)
        ws.append(w)
    if sorted(w.tile_shape for w in ws) == [0, 2, 3]:
        arr = (C.c_void_p * len(ws))(*[w.h for w in ws])
        with pytest.raises(_lib.Rem2dError, match="shapes 0 and 2"):
            _lib.check(_lib.lib().rem2d_worlds_step(arr, len(ws), 1, ws[0]._stream()))
        sg = (_lib.StepGroup * 1)()
        sg[0].worlds, sg[0].n_worlds, sg[0].stream = arr, len(ws), None
        with pytest.raises(_lib.Rem2dError, match="shapes 0 and 2"):
            _lib.check(_lib.lib().rem2d_groups_step(sg, 1, 1, ws[0]._stream(), 0))
        with pytest.raises(_lib.Rem2dError, match="shapes 0 and 2"):
            _lib.check(_lib.lib().rem2d_worlds_launch_info(arr, len(ws), None, None))
        # ... and the same three worlds step fine as two launches
        _lib.check(_lib.lib().rem2d_worlds_step(arr, 2, 5, ws[0]._stream()))
        _lib.check(_lib.lib().rem2d_world_step(ws[2].h, 5, ws[0]._stream()))
        for w, m in zip(ws, morphs):
            ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), m.as_dict(), 5, n_threads=8, flags=oracle.FLAG_CONTINUOUS)
            assert np.array_equal(w.bodies(), ref["bodies"]) and int(w.view("err").max()) == 0
    for w in ws:
        w.close()
    w = gpu(morphs[0].n_envs, morphs[0].lanes)
    with pytest.raises(_lib.Rem2dError, match="tile shape"):
        _lib.check(_lib.lib().rem2d_world_set_tile_shape(w.h, 5))
    w.close()


def test_fma_tolerance_mode_transition_parity(gpu, rough_terrain):
    """The labelled TOLERANCE MODE (gym_rem2d_amd/librem2d_fma.so: the same source built with -ffp-contract=fast, +7 % env-steps/s)
    is NOT the bit-exact product -- north_star allows "a stated fp32 tolerance on positions/velocities", and this test is where it
    is stated.  SURVEY.md 8c's protocol (i), transition parity: identical full state in (the strict world's arena, copied before
    every step), one env-step out, > 10^5 transitions that include the landing (TOI events), tools/fma_tolerance.py:
      * the integer state (awake, limit states, pair lists, manifold point counts / types, feature keys, done, TOI events) is equal
        in >= 99.98 % of the transitions (an ulp can flip a comparison: measured 5 of 153 600);
      * where it is equal, poses are within 1e-5 + 1e-4 |x| in >= 99.95 % (measured 99.99 %; worst 2.3e-3) and velocities in
        >= 99 % (measured 99.8 %; worst 0.024) -- one step of a 180-sweep Gauss-Seidel solve amplifies an ulp, chaos does the rest
        along a trajectory, which is why trajectories are never compared;
      * accumulated contact impulses are NOT held to it (the two points of a block solve share their load ill-conditionedly:
        2-3 % of the transitions beyond the tolerance, sums unaffected) -- they are warm-start values, not observables.
    The strict build stays the default, the headline and the only thing the committed digests guard."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fma_tolerance as FT
    from gym_rem2d_amd import _lib
    assert _lib.build_id("fma") != _lib.build_id() and _lib.build_id("fma") == _lib.source_id(_lib.FMA_FLAGS)
    m = FT.default_population(768)
    r = FT.transition_parity(m, rough_terrain, 140, _lib.FLAG_CONTINUOUS)
    t = r["transitions"]
    assert t >= 100000 and r["toi_event_transitions"] > 500 and r["errors"] == 0
    assert r["bit_identical_creatures"] < t            # (it IS a different build: not everything can agree bit for bit)
    assert r["int_mismatch_creatures"] <= 2e-4 * t, r
    by = r["by_field"]
    assert by["px"][0] + by["py"][0] + by["ang"][0] <= 5e-4 * t, r
    assert by["vx"][0] + by["vy"][0] + by["w"][0] <= 1e-2 * t, r
    assert max(by["px"][1], by["py"][1], by["ang"][1]) < 0.02 and max(by["vx"][1], by["vy"][1], by["w"][1]) < 0.5, r
