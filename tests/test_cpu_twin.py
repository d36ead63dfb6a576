"""oracle/librem2d_cpu.so: include/rem2d.h's ABI on host pointers (the `rem2d_cpu_*` twin SURVEY.md 8b proposed).

CPU tests: the twin exports a counterpart of every world entry point of include/rem2d.h, sizes the arena like the HIP
library, and steps exactly like the oracle's own batch API.  GPU test: ONE call sequence through both libraries, then the
two arenas compared field by field."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# entry points of include/rem2d.h without a twin, and why
HOST_ONLY = {
    "rem2d_plan_tiles", "rem2d_plan_tiles_shape", "rem2d_compile_lsystem", "rem2d_compile_trees", "rem2d_compile_network",  # host code already
    "rem2d_mutate_trees",
    "rem2d_tree_diversity",  # its CPU restatement is oracle.tree_distance_matrix (tests/test_diversity.py)
    "rem2d_selftest_scalar",  # its CPU restatement is rem2d_oracle_kat_scalar / rem2d_oracle_sincosf
}


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    from gym_rem2d_amd.world import BatchedWorld
    return BatchedWorld


def _twin():
    from oracle import cpu_twin
    cpu_twin.build()
    return cpu_twin


def _declared():
    text = open(os.path.join(ROOT, "include", "rem2d.h")).read()
    return sorted(set(re.findall(r"\b(rem2d_[a-z_0-9]+)\s*\(", text)))


def test_twin_exports_every_world_entry_point():
    T = _twin()
    L = C.CDLL(T._SO)
    missing = []
    for name in _declared():
        if name in HOST_ONLY:
            continue
        twin = name.replace("rem2d_", "rem2d_cpu_", 1)
        if not hasattr(L, twin):
            missing.append(twin)
    assert not missing, missing
    assert L.rem2d_cpu_abi_version() == int(re.search(r"#define REM2D_ABI_VERSION (\d+)",
                                                      open(os.path.join(ROOT, "include", "rem2d.h")).read()).group(1))


def test_twin_sizes_the_arena_like_the_hip_library():
    """rem2d_state_bytes / rem2d_padded_envs are host code in librem2d.so: callable without a GPU."""
    from gym_rem2d_amd import _lib
    T = _twin()
    _lib.build()
    L = _lib.lib()  # imports torch first: librem2d.so must bind to the HIP runtime torch ships (see _lib.lib)
    T.lib()
    for n_envs in (1, 3, 16, 33, 1000, 65536):
        for lanes in (2, 4, 8, 16, 32, 64):
            a, b = _lib.WorldCfg(n_envs, lanes, 0, 0), T.WorldCfg(n_envs, lanes, 0, 0)
            assert L.rem2d_state_bytes(C.byref(a)) == T.lib().rem2d_cpu_state_bytes(C.byref(b))
            assert L.rem2d_padded_envs(C.byref(a)) == T.lib().rem2d_cpu_padded_envs(C.byref(b))
    for bad in ((0, 4), (4, 3), (4, 128)):
        assert T.lib().rem2d_cpu_state_bytes(C.byref(T.WorldCfg(bad[0], bad[1], 0, 0))) == 0


def _population():
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology, lanes_for
    specs = synthetic.lsystem_specs(range(40))
    groups = {}
    for s in specs:
        groups.setdefault(lanes_for(s.n_bodies), []).append(s)
    k = max(groups, key=lambda q: len(groups[q]))
    return Morphology.from_specs(groups[k], k)


def test_twin_steps_like_the_oracle_batch_api():
    from oracle import oracle as O
    from gym_rem2d_amd import make_terrain
    T = _twin()
    morph = _population()
    terrain = make_terrain(4)
    xs, ys, polys = terrain.f32()
    ot = O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
    for flags in (0, 1):
        ref = O.batch_run(ot, morph.as_dict(), 150, n_threads=4, flags=flags)
        w = T.CpuWorld(morph.n_envs, morph.lanes, flags)
        w.set_terrain(terrain)
        w.reset(morph)
        assert np.array_equal(w.view("px"), morph.arrays["x"].reshape(morph.n_envs, morph.lanes) * (w.view("shape") != 0))
        for n in (1, 49, 100):
            w.step(n)
        assert np.array_equal(w.bodies(), ref["bodies"])
        assert np.array_equal(w.view("reward"), ref["reward"].astype(np.float32))
        assert np.array_equal(w.view("everdone"), ref["done"])
        assert np.array_equal(w.view("fitness"), ref["fitness"])
        assert int(w.view("steps").min()) == 150
        w.close()


def test_twin_call_order_and_argument_errors():
    T = _twin()
    morph = _population()
    w = T.CpuWorld(morph.n_envs, morph.lanes)
    with pytest.raises(T.CpuTwinError, match="set_terrain"):
        w.reset(morph)
    with pytest.raises(T.CpuTwinError, match="precede step"):
        w.step(1)
    with pytest.raises(T.CpuTwinError):
        T.CpuWorld(0, 4)
    w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flags,steps", [(0, (0, 1, 30, 120)), (1, (0, 1, 30, 120)), (1 | 2, (40, 160)), (1 | 4, (40, 160)),
                                         (1 | 8, (60, 90, 150, 100)), (1 | 8 | 16, (60, 90, 150, 100))])
def test_same_call_sequence_same_arena(gpu, flags, steps):
    """The drop-in statement at the boundary: one sequence of ABI calls, two libraries, equal state fields -- discrete
    and continuous physics, both sleep variants, and REM2D_FLAG_SKIP_FROZEN (8: wavefronts whose creatures all have a
    final fitness stop being stepped, in both libraries at the same step), the latter also with REM2D_FLAG_RETILE (16: a launch
    shape of the HIP library that the twin ignores -- the arenas must not show it)."""
    from gym_rem2d_amd import make_terrain
    from gym_rem2d_amd.world import BatchedWorld
    T = _twin()
    morph = _population()
    terrain = make_terrain(4)
    g = BatchedWorld(morph.n_envs, morph.lanes, flags=flags)
    c = T.CpuWorld(morph.n_envs, morph.lanes, flags)
    for w in (g, c):
        w.set_terrain(terrain)
        w.reset(morph)
    # same layout
    for name in T.FIELDS:
        off, cnt, dt = C.c_size_t(), C.c_size_t(), C.c_int32()
        from gym_rem2d_amd import _lib
        _lib.check(_lib.lib().rem2d_world_field(g.h, _lib.FIELD_ID[name], C.byref(off), C.byref(cnt), C.byref(dt)))
        assert (off.value, cnt.value, dt.value) == c.field(name), name
    active = c.view("shape") != 0
    jointed = active & (c.view("parent") >= 0)
    lane_fields = ["px", "py", "ang", "vx", "vy", "w", "sleept", "hx", "hy", "invm", "invi", "fatlx", "fatly", "fatux",
                   "fatuy", "awake", "ccount", "shape"]
    joint_fields = ["jax", "jay", "jbx", "jby", "jtorque", "jlower", "jupper", "jimpx", "jimpy", "jimpz", "jmotorimp",
                    "jmotorspeed", "jlimit", "camp", "cphase", "cfreq", "coffset", "cistate", "parent"]
    env_fields = ["wod", "fitness", "reward", "done", "everdone", "frozen", "steps", "invdt0", "positers", "toievents"]
    total = 0
    for n in steps:
        if n:
            g.step(n)
            c.step(n)
            total += n
        for name in lane_fields:
            assert np.array_equal(g.view(name).cpu().numpy()[active], c.view(name)[active]), (name, total)
        for name in joint_fields:
            assert np.array_equal(g.view(name).cpu().numpy()[jointed], c.view(name)[jointed]), (name, total)
        for name in env_fields:
            assert np.array_equal(g.view(name).cpu().numpy(), c.view(name)), (name, total)
        # pair lists in list order: edge, point count and manifold type, feature keys and impulses of the points
        cc = c.view("ccount")
        gi, ci = g.view("cinfo").cpu().numpy(), c.view("cinfo")
        for k in range(int(cc.max())):
            m = active & (cc > k)
            assert np.array_equal(g.view("cedge").cpu().numpy()[k][m], c.view("cedge")[k][m]), ("cedge", total)
            assert np.array_equal(gi[k][m] & 0xff, ci[k][m] & 0xff), ("point count", total)
            touching = m & ((ci[k] & 0xff) > 0)
            assert np.array_equal((gi[k][touching] >> 8) & 0xff, (ci[k][touching] >> 8) & 0xff), ("manifold type", total)
            for j, (key, nn, tt) in enumerate((("ckey0", "cn0", "ct0"), ("ckey1", "cn1", "ct1"))):
                mj = m & ((ci[k] & 0xff) > j)
                for name in (key, nn, tt):
                    assert np.array_equal(g.view(name).cpu().numpy()[k][mj], c.view(name)[k][mj]), (name, total)
    assert int(g.view("err").max()) == 0
    if flags & 8:
        assert 0 < int(c.view("frozen").sum()) and int(c.view("steps").min()) < total  # some wavefronts did stop early
    g.close()
    c.close()


@pytest.mark.gpu
def test_step_ex_and_outputs_through_both_libraries(gpu):
    """rem2d_world_step_ex with other iteration counts / dt, and rem2d_world_set_outputs (population-order reward/done)."""
    import torch
    from gym_rem2d_amd import make_terrain
    from gym_rem2d_amd.world import BatchedWorld
    T = _twin()
    morph = _population()
    terrain = make_terrain(4, flat=True)
    g = BatchedWorld(morph.n_envs, morph.lanes, flags=1)
    c = T.CpuWorld(morph.n_envs, morph.lanes, 1)
    perm = np.random.RandomState(3).permutation(morph.n_envs).astype(np.int32)
    g_out = (torch.zeros(morph.n_envs, device="cuda"), torch.zeros(morph.n_envs, dtype=torch.bool, device="cuda"),
             torch.from_numpy(perm).cuda())
    c_out = (np.zeros(morph.n_envs, np.float32), np.zeros(morph.n_envs, np.uint8), perm.copy())
    for w, o in ((g, g_out), (c, c_out)):
        w.set_terrain(terrain)
        w.reset(morph)
        w.set_outputs(*o)
        w.step_ex(25, 1.0 / 50.0, 180, 60)
        w.step_ex(10, 1.0 / 60.0, 8, 3)
        w.step_ex(15, 1.0 / 50.0, 30, 0)
        w.step(5)
    active = c.view("shape") != 0
    for name in ("px", "py", "ang", "vx", "vy", "w", "sleept", "awake", "jimpx", "jimpy", "jimpz", "jmotorimp"):
        assert np.array_equal(g.view(name).cpu().numpy()[active], c.view(name)[active]), name
    for name in ("wod", "fitness", "reward", "done", "steps", "invdt0", "positers", "toievents"):
        assert np.array_equal(g.view(name).cpu().numpy(), c.view(name)), name
    assert np.array_equal(g_out[0].cpu().numpy(), c_out[0])
    assert np.array_equal(g_out[1].cpu().numpy().astype(np.uint8), c_out[1])
    assert np.array_equal(c_out[0][perm], c.view("reward"))
    g.close()
    c.close()
