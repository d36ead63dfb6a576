"""The step train's hand-over check end to end (pytest -m gpu): REM2D_ERR_HANDOVER is produced, seen, told apart from a
contact-capacity overflow and recovered from.

Modular2DEnv.py:634 is one synchronous ``world.Step``: one result whatever the launch form.  The step train
(rem2d_step_train_kernel, the default launch form) hands a block from the workgroup of step s to the workgroup of step s + 1; a
hand-over it cannot vouch for (published from another XCD, or never seen within 2 s) flags the block's creatures instead of
handing out their state.  ``REM2D_OPT_TRAIN_FAULT`` (a test hook, include/rem2d.h) makes chosen (step, block) workgroups report
exactly that -- or really withholds a flag -- without touching the arithmetic.
"""
import warnings

import numpy as np
import pytest

from conftest import oracle_terrain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def need_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()


def _specs(n=192):
    from gym_rem2d_amd import synthetic
    return synthetic.lsystem_specs(range(n), mutate_odd=True)


def test_forced_handover_failure_is_flagged_named_and_recovered(need_gpu, oracle, rough_terrain):
    """Workgroups (step 3 of every launch, every 2nd block) are told that their hand-over failed.  (1) the creatures of exactly
    those blocks carry REM2D_ERR_HANDOVER and the host-side counter moves; (2) BatchedModular2D.step / .fitness raise
    HandoverError -- the message names the hand-over, not the solver slots; (3) run_episode evaluates the flagged creatures again
    on per-step launches of the SAME build (never the wide one) and every fitness == the oracle's; (4) the strict modes raise
    HandoverError, the penalty mode never scores a hand-over as UNRESOLVED_FITNESS."""
    import torch
    from gym_rem2d_amd import Morphology, _lib
    from gym_rem2d_amd.env import BatchedModular2D
    from gym_rem2d_amd.evaluate import HandoverError, SolverOverflow, run_episode
    specs = _specs()
    ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), Morphology.from_specs(specs, 16).as_dict(), 2500, n_threads=8,
                           flags=oracle.FLAG_CONTINUOUS)["fitness"]
    fault = {"train_fault": _lib.train_fault(step=3, every_block=2)}

    # (1) + (2): the facade does not hand out anything computed from a flagged state silently
    env = BatchedModular2D(seed=4, options=fault, on_handover="flag")
    env.reset_specs(specs)
    assert env.launch_info()[1] == 2, "the default launch form must be the step train"
    env.step(10)                              # ("flag": nothing raises here, whether the launch has finished or not)
    torch.cuda.synchronize()
    assert env.handover_failures() > 0
    env.on_handover = "raise"
    err = env.errors().cpu().numpy()
    assert ((err & _lib.ERR_HANDOVER) != 0).any() and not ((err & _lib.ERR_HANDOVER) != 0).all() and not (err & _lib.ERR_CAPACITY).any()
    with pytest.raises(HandoverError, match="hand-over") as ei:
        env.step(1)
    assert ei.value.failures == env.handover_failures() and "solver / pair slots" not in str(ei.value)
    assert not isinstance(ei.value, SolverOverflow)
    with pytest.raises(HandoverError):
        env.fitness
    # which creatures: exactly those of every 2nd 64-lane block of the train's launch order -- per world, blocks are numbered over
    # the group's worlds (widest lane bucket last); at least: whole blocks, i.e. creatures sharing a block share the verdict
    for w, idx in env.worlds:
        e = (w.view("err").cpu().numpy() & _lib.ERR_HANDOVER) != 0
        per_block = 64 // w.lanes
        blocks = e[:len(e) // per_block * per_block].reshape(-1, per_block)
        assert (blocks.all(axis=1) | ~blocks.any(axis=1)).all()
    env.close()

    # (3) the episode path recovers: same build, per-step launches, oracle's fitness for everybody
    env = BatchedModular2D(seed=4, options=fault)
    env.reset_specs(specs)
    with pytest.warns(UserWarning, match="REM2D_ERR_HANDOVER"):
        fit = run_episode(env, max_steps=2500)
    assert np.array_equal(fit.cpu().numpy(), ref)
    assert len(env.last_handover) > 0 and env.last_overflow == [] and env.last_unresolved == []
    # (4) strict modes name it; penalty mode resolves it instead of penalising it
    env.reset_specs(specs)
    with pytest.raises(HandoverError):
        run_episode(env, max_steps=200, on_error="raise")
    env.reset_specs(specs)
    env.handover_failures(clear=True)
    with pytest.warns(UserWarning, match="REM2D_ERR_HANDOVER"):
        fit = run_episode(env, max_steps=2500, on_error="penalty")
    assert np.array_equal(fit.cpu().numpy(), ref) and env.last_unresolved == []
    env.close()

    # and without the fault: no warning, no flag, the same fitness
    env = BatchedModular2D(seed=4)
    env.reset_specs(specs)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        fit = run_episode(env, max_steps=2500)
    assert np.array_equal(fit.cpu().numpy(), ref) and env.last_handover == [] and env.handover_failures() == 0
    env.close()


def test_withheld_flag_times_out_flags_and_drains(need_gpu):
    """The real mechanism: the flag of (step 1, every block) of a launch is never published.  The waiters of step 2 run into the
    2 s limit, the launch DRAINS (flags[1]: every later wait ends at once -- a 40-step launch takes seconds, not 40 x 2 s), every
    creature the launch touched afterwards carries REM2D_ERR_HANDOVER, the counter moves, nothing hangs."""
    import time
    import torch
    from gym_rem2d_amd import _lib, make_terrain, synthetic
    from gym_rem2d_amd.world import BatchedWorld
    m = synthetic.chain_population(256, 4, "left")
    w = BatchedWorld(m.n_envs, m.lanes, flags=_lib.FLAG_CONTINUOUS, options={"train_fault": _lib.train_fault(2, 1, drop=True)})
    w.set_terrain(make_terrain(4))
    w.reset(m)
    t0 = time.time()
    w.step(40)
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert 1.5 < dt < 20.0, dt
    assert w.handover_failures() > 0
    assert bool(((w.view("err") & _lib.ERR_HANDOVER) != 0).all())
    assert w.handover_failures(clear=True) > 0 and w.handover_failures() == 0
    w.close()


def test_long_launch_step_count_field(need_gpu):
    """the flag used to hold `steps done` in 16 bits, so a call of >= 65 536 steps on a small world could never be
    satisfied.  The field is 28 bits now and a launch is cut at its capacity: 70 000 steps in ONE call (cheap iterations: the
    point is the count) == the same steps in two calls, no hand-over failure."""
    import torch
    from gym_rem2d_amd import _lib, make_terrain, synthetic
    from gym_rem2d_amd.world import BatchedWorld
    m = synthetic.chain_population(32, 2, "left")
    outs = []
    for calls in ((70000,), (35000, 35000)):
        w = BatchedWorld(m.n_envs, m.lanes, flags=0)
        w.set_terrain(make_terrain(4, flat=True))
        w.reset(m)
        for n in calls:
            w.step_ex(n, 1.0 / 50, 1, 1)
        torch.cuda.synchronize()
        assert w.handover_failures() == 0 and int(w.view("err").max()) == 0
        assert int(w.view("steps").min()) == 70000
        outs.append(w.bodies())
        w.close()
    assert np.array_equal(outs[0], outs[1])


def test_step_time_counts_env_steps(need_gpu):
    """with the step train one time bracket spans a whole launch; rem2d_world_step_time_ms reports ENV-STEPS."""
    import torch
    from gym_rem2d_amd import _lib, make_terrain, synthetic
    from gym_rem2d_amd.world import BatchedWorld
    m = synthetic.chain_population(128, 4, "left")
    w = BatchedWorld(m.n_envs, m.lanes, flags=_lib.FLAG_CONTINUOUS)
    w.set_terrain(make_terrain(4))
    w.reset(m)
    w.enable_timing(True)
    w.step(25)
    w.step(5)
    ms, n = w.step_time_ms()
    kms, launches = w.kernel_time_ms()
    assert n == 30 and launches == 2 and 0 < kms <= ms
    w.set_option("fuse_velpost", 1)
    w.step(7)
    ms, n = w.step_time_ms()
    assert n == 7 and ms > 0
    w.close()


@pytest.mark.parametrize("shape,uniform", [(1, False), (4, True), (4, False)])
def test_train128_is_the_launch_form_and_matches_the_oracle(need_gpu, oracle, rough_terrain, shape, uniform):
    """The 128-lane tile shapes have a step train of their own (rem2d_step_train128_kernel: an item = a tile's one or two blocks).
    rem2d_worlds_launch_info names it (2), its result == the oracle's == per-step launches of the same shape, a forced hand-over
    failure is flagged there too -- and the STATIC shape under a creature order keeps per-step launches (its velocity tiles hold
    the arena order while pre / post follow the creature order: legal between launches, not inside one workgroup)."""
    import ctypes as C
    import torch
    from gym_rem2d_amd import Morphology, _lib, synthetic
    from gym_rem2d_amd.world import BatchedWorld
    if uniform:
        morph = synthetic.chain_population(96, 8, "left")
    else:
        specs = [s for s in synthetic.lsystem_specs(range(300), mutate_odd=True) if 3 <= s.n_bodies <= 8]
        morph = Morphology.from_specs(specs, 8)
    ref = oracle.batch_run(oracle_terrain(oracle, rough_terrain), morph.as_dict(), 150, n_threads=8, flags=oracle.FLAG_CONTINUOUS)

    def info(w):
        arr = (C.c_void_p * 1)(w.h)
        s, f = C.c_int32(), C.c_int32()
        _lib.check(w.L.rem2d_worlds_launch_info(arr, 1, C.byref(s), C.byref(f)))
        return s.value, f.value
    outs = []
    for opts in (None, {"fuse_velpost": 1}, {"train_fault": _lib.train_fault(2, 2)}):
        w = BatchedWorld(morph.n_envs, morph.lanes, _lib.FLAG_CONTINUOUS, options=opts)
        w.set_terrain(rough_terrain)
        w.reset(morph, tile_shape=shape)
        regular = info(w)
        assert regular == (shape, 0 if opts and "fuse_velpost" in opts else 2), regular
        for n in (1, 49, 100):
            w.step(n)
        torch.cuda.synchronize()
        outs.append(w.bodies())
        assert np.array_equal(outs[-1], ref["bodies"])          # (a forced failure changes flags, never arithmetic)
        assert np.array_equal(w.view("fitness").cpu().numpy(), ref["fitness"])
        err = w.view("err").cpu().numpy()
        if opts and "train_fault" in opts:
            assert w.handover_failures() > 0 and ((err & _lib.ERR_HANDOVER) != 0).any() and not (err & _lib.ERR_CAPACITY).any()
        else:
            assert w.handover_failures() == 0 and int(err.max()) == 0
        if opts is None and shape == 4:
            # a creature order on the static shape: the library answers with per-step launches from here on, same bits
            w.set_order(torch.randperm(morph.n_envs, generator=torch.Generator().manual_seed(3)))
            assert info(w) == (4, 0)
        w.close()


def test_back_to_back_handovers_small_population(need_gpu):
    """The hand-over under stress: with a few hundred creatures the workgroup of a block's next step is dispatched right behind the one
    it waits for (a big population keeps them a whole step of other items apart), so every step really polls a flag that is being
    published.  The train against per-step launches, every arena field `==`, through 250 calls of random lengths (the same check over
    21 000 launches: profiles/r06_train_vs_steps_soak_small.jsonl)."""
    import torch
    from gym_rem2d_amd import _lib, synthetic
    from gym_rem2d_amd.env import BatchedModular2D
    rng = np.random.default_rng(11)
    batches = [(m, idx.tolist()) for m, idx in synthetic.lsystem_batches_native(range(5_000_000, 5_000_256), n_proc=1)]
    envs = []
    for fuse in (2, 1):
        e = BatchedModular2D(seed=4, flags=_lib.FLAG_CONTINUOUS, options={"fuse_velpost": fuse})
        e.rebalance_every, e.step_groups = 50, 1
        e._upload(batches, 256)
        envs.append(e)
    assert envs[0].launch_info()[1] == 2 and envs[1].launch_info()[1] == 1
    for call in range(250):
        n = int(rng.integers(1, 25))
        for e in envs:
            e.step(n)
        if call % 50 == 49:
            torch.cuda.synchronize()
            for (wa, _), (wb, _) in zip(envs[0].worlds, envs[1].worlds):
                for name in _lib.FIELDS:
                    a, b = wa.view(name), wb.view(name)
                    assert torch.equal(a, b) or (a.dtype.is_floating_point and bool((a == b).all())), (name, wa.lanes, call)
    assert envs[0].handover_failures() == 0 and int(envs[0].errors().max()) == 0
    for e in envs:
        e.close()
