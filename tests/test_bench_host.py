"""Host-side pieces of bench.py that need no GPU: the self-launcher's command line and the CPU-baseline leg."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_command(monkeypatch):
    import subprocess
    import bench
    seen = {}

    class Done(Exception):
        pass

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw

        class P:
            returncode = 0
            stdout = 'noise\n{"n_gpus": 2}\n'
        return P()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.setattr(sys, "exit", lambda code=0: (_ for _ in ()).throw(Done(code)))
    ap_args = type("A", (), {"gpus": 2})()
    try:
        bench.launch_ranks(ap_args)
    except Done as e:
        assert e.args[0] == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_cpu_baseline_times_the_gpu_legs_window(oracle):
    """The oracle leg runs ONE batch over all lane buckets, over the step window [settle, settle + window) -- not the
    free fall right after reset -- and reports all-thread and one-thread rates."""
    import bench
    from gym_rem2d_amd import make_terrain, synthetic
    from gym_rem2d_amd.compiler import Morphology, lanes_for
    groups = {}
    for s in synthetic.lsystem_specs(range(48)):
        groups.setdefault(lanes_for(s.n_bodies), []).append(s)
    morphs = [Morphology.from_specs(groups[k], k) for k in sorted(groups)]
    cb = bench.cpu_baseline(morphs, make_terrain(4, flat=True), 1, settle=30, window=25, budget_s=5.0)
    assert cb["kind"] == "port" and cb["is_oracle"] and cb["unit"] == "env-steps/s"
    assert cb["cores"] == bench.host_cores()[0] <= len(os.sched_getaffinity(0))
    assert cb["value"] > 0 and cb["value_1thread"] > 0
    assert "steps [30, " in cb["sample"] and "48 creatures" in cb["sample"] and "one continuous timed window" in cb["sample"]
    json.dumps(cb)
    # the re-laid-out batch is the same physics: 16-lane repack of a 4-lane bucket gives the oracle the same bodies
    m = morphs[0]
    xs, ys, _ = make_terrain(4, flat=True).f32()
    ot = oracle.Terrain(xs, ys, None, make_terrain(4, flat=True).friction)
    a = oracle.batch_run(ot, m.as_dict(), 40, n_threads=2, flags=1)["bodies"]
    b = oracle.batch_run(ot, bench._repack(m, 16), 40, n_threads=2, flags=1)["bodies"]
    assert np.array_equal(a, b[:, :m.lanes]) and not b[:, m.lanes:].any()


def test_timed_blocks_protocol():
    """Blocks of exactly K steps between a sync on both sides, repeated until the timed region reaches min_time (at least
    one block, at most max_blocks); the block time every rank sees is the reduced (max over ranks) one."""
    import bench
    calls = {"sync": 0, "steps": []}
    clock = iter(range(1000))

    def fake_block(n):
        calls["steps"].append(n)

    def sync():
        calls["sync"] += 1
    blocks = bench.timed_blocks(fake_block, 20, 0.0, 10, sync, lambda x: 0.25)
    assert blocks == [0.25] and calls["steps"] == [20] and calls["sync"] == 2
    calls.update(sync=0, steps=[])
    blocks = bench.timed_blocks(fake_block, 7, 1.0, 100, sync, lambda x: 0.3)
    assert blocks == [0.3] * 4 and calls["steps"] == [7] * 4 and calls["sync"] == 8     # 4 x 0.3 s >= 1 s
    blocks = bench.timed_blocks(fake_block, 7, 1e9, 5, sync, lambda x: 0.3)
    assert len(blocks) == 5                                                             # capped
    cores, quota = bench.host_cores()
    assert 1 <= cores <= (os.cpu_count() or 1) and (quota is None or quota > 0)
    assert bench.METRIC == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]


def test_bench_population_is_the_seeded_python_population(tmp_path, monkeypatch):
    """bench.py builds its populations with the native compilers (genomes by seed in Python, everything after natively):
    the lane buckets must be, word for word and in the same order, what Morphology.from_specs gives for
    synthetic.lsystem_specs / cppn_specs of the same seeds sorted the way BatchedModular2D.reset_specs sorts them."""
    import tempfile
    import bench
    from gym_rem2d_amd import synthetic
    from gym_rem2d_amd.compiler import Morphology, lanes_for
    monkeypatch.setattr(tempfile, "gettempdir", lambda: str(tmp_path))     # (no stale genome cache, none left behind)
    monkeypatch.setenv("REM2D_BENCH_NO_FORK", "1")
    for workload, specs_fn, first, n in (("lsystem", synthetic.lsystem_specs, 4000, 700), ("cppn_hardcore", synthetic.cppn_specs, 50, 300)):
        for _ in range(2):                                                 # second pass: from the on-disk genome cache
            morphs, desc = bench.finish_population(bench.build_population(workload, n, first))
            specs = specs_fn(range(first, first + n))
            groups = {}
            for sp in specs:
                groups.setdefault(lanes_for(sp.n_bodies), []).append(sp)
            assert [m.lanes for m in morphs] == sorted(groups)
            for m in morphs:
                g = sorted(groups[m.lanes], key=lambda sp: (sp.period, max(sp.rounds, default=-1), sp.n_bodies), reverse=True)
                ref = Morphology.from_specs(g, m.lanes)
                assert np.array_equal(ref.n_bodies, m.n_bodies)
                for k in ref.arrays:
                    assert np.array_equal(ref.arrays[k], m.arrays[k]), (workload, m.lanes, k)
            assert "seeds %d..%d" % (first, first + n - 1) in desc
    assert sorted(os.listdir(tmp_path)) == ["rem2d_bench_genomes_cppn_hardcore_300_50.npz", "rem2d_bench_genomes_lsystem_700_4000.npz"]


def test_cpu_baseline_sample_is_capped_by_host_memory(monkeypatch, oracle):
    """The oracle worlds of the CPU-baseline window are alive at once (~0.6 MB each): the sample is capped by a quarter of the
    memory the cgroup / the machine still has, so that a small container does not run out of memory after the GPU result is
    in."""
    import bench
    from gym_rem2d_amd import make_terrain, synthetic
    from gym_rem2d_amd.compiler import Morphology
    b = bench.host_memory_budget()
    assert 0 < b <= 4 << 30
    monkeypatch.setattr(bench, "host_memory_budget", lambda: 70 * bench.ORACLE_WORLD_BYTES)
    m = Morphology.from_specs(synthetic.lsystem_specs(range(400)), 16)
    cb = bench.cpu_baseline([m], make_terrain(4, flat=True), 1, settle=5, window=20, budget_s=2.0)
    n = int(cb["sample"].split(" creatures")[0])
    assert 32 <= n <= 70, cb["sample"]


def test_reference_baseline_leg_is_dormant_without_the_wheel(monkeypatch):
    """`try: import Box2D` -> the reference's own step loop as `cpu_baseline.kind = "reference"`; else the port.  The wheel
    (Box2D==2.3.10, the reference's requirements.txt:1) is on neither box: the leg answers None and never looks for a checkout --
    and even WITH the wheel it reads a reference checkout only where REM2D_REFERENCE names one (never /root/reference by default:
    that path does not exist on the GPU box)."""
    import bench
    try:
        import Box2D  # noqa: F401
        have = True
    except ImportError:
        have = False
    monkeypatch.delenv("REM2D_REFERENCE", raising=False)
    assert bench.reference_baseline(25, 100) is None
    if not have:
        monkeypatch.setenv("REM2D_REFERENCE", "/root/reference/ModularER_2D")
        assert bench.reference_baseline(25, 100) is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "/root/reference" not in src.split("def reference_baseline")[1].split("\ndef ")[0].replace(
        "nothing is\n    read from /root/reference by default", "")
    assert bench.REFERENCE_PYTHON_OVERHEAD["step_us_color_control_on"] == 2287


def test_scaling_claim_names_what_the_6x_is_judged_on():
    """the line of a multi-rank run says which scaling north_star's ">= 6x at 8 GPUs" is judged on (weak: 65 536 per
    GPU), carries the committed single-GPU line it is compared with and the ratio, and labels a strong-scaling line as such."""
    import bench
    ref = bench.n1_reference()
    assert ref is not None and ref["source"].startswith("profiles/r06_") and 5e7 < ref["value"] < 1e8
    weak = bench.scaling_claim(8, False, 8 * 65536, 65536, 8 * ref["value"])
    assert weak["this_line"] == "weak" and weak["creatures_total"] == 524288 and weak["creatures_per_gpu"] == 65536
    assert abs(weak["vs_n1"] - 8.0) < 1e-9 and "weak scaling" in weak["judged_on"] and ">= 6x" in weak["north_star"]
    strong = bench.scaling_claim(8, True, 65536, 8192, 1.2 * ref["value"])
    assert strong["this_line"] == "strong" and "1.2x" in strong["expected"] and abs(strong["vs_n1"] - 1.2) < 1e-9
    # and the roofline's hardware ceiling is the guide's figure, the measured ones sit beside it
    assert bench.HW_VALU_ISSUE_PEAK == 1024 * 2.4e9 / 2
    assert bench.UBENCH_INDEPENDENT["wave_instructions_per_s_at_4_waves_per_simd"] < bench.UBENCH_INDEPENDENT["wave_instructions_per_s_at_8_waves_per_simd"] < bench.HW_VALU_ISSUE_PEAK
