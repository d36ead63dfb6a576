"""ctypes binding of the C ABI in include/rem2d.h (librem2d.so, HIP/gfx950).

The product path has no CPU fallback: if the library is missing this raises, loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("REM2D_LIB_PATH") or os.path.join(_HERE, "librem2d.so")   # override: A/B builds on the GPU box
# the same source built with -DREM2D_WIDE (32 pair slots / 12 solver slots per body): where the creatures that overflowed the
# default build's slots are re-evaluated (evaluate.run_episode) -- Box2D itself has no cap (Modular2DEnv.py:634)
WIDE_LIB_PATH = os.environ.get("REM2D_WIDE_LIB_PATH") or os.path.join(_HERE, "librem2d_wide.so")
# the same source built with -ffp-contract=fast: the compiler fuses a * b + c into v_fma_f32 wherever it likes.  NOT the bit-exact
# product: a labelled TOLERANCE MODE (north_star: "within a stated fp32 tolerance on positions/velocities") that
# tests/test_parity_gpu.py validates transition by transition against the strict build and bench.py reports as a secondary line
# beside the strict headline.  Selected with wide="fma" wherever a `wide` argument is taken.
FMA_LIB_PATH = os.environ.get("REM2D_FMA_LIB_PATH") or os.path.join(_HERE, "librem2d_fma.so")
SRC_PATH = os.path.join(_HERE, "csrc", "rem2d.hip")

FLAG_CONTINUOUS = 1
FLAG_SLEEP_RESET_ALWAYS = 2
FLAG_NO_SLEEP = 4
FLAG_SKIP_FROZEN = 8
FLAG_RETILE = 16

CONTACT_SLOTS = 24
MAX_WORLDS_PER_STEP = 8
MAX_STEP_GROUPS = 16
STEP_GRAPH = 1
SOLVER_SLOTS = 6
ERR_PAIR_OVERFLOW = 1
ERR_SOLVER_OVERFLOW = 2
ERR_HANDOVER = 4   # (the step train's hand-over check, include/rem2d.h)
ERR_CAPACITY = ERR_PAIR_OVERFLOW | ERR_SOLVER_OVERFLOW


def train_fault(step, every_block=1, drop=False):
    """Value of the `train_fault` test option (REM2D_OPT_TRAIN_FAULT): the workgroups of step `step` (1-based inside a launch) of
    the blocks with block % every_block == 0 are told that their hand-over failed / (drop) never get their flag."""
    return int(step) | (int(every_block) << 16) | ((1 << 30) if drop else 0)

# launch options of a world (include/rem2d.h REM2D_OPT_*, rem2d_world_set_option): no result depends on them
OPTIONS = ("pipeline", "fuse_velpost", "prio", "prio_t1", "prio_t2", "heavy_per_wave", "debug", "rebalance", "train_fault")
OPTION_ID = {n: i for i, n in enumerate(OPTIONS)}
_ENV_OPTIONS = {"REM2D_PIPELINE": "pipeline", "REM2D_FUSE_VELPOST": "fuse_velpost", "REM2D_PRIO": "prio",
                "REM2D_PRIO_T1": "prio_t1", "REM2D_PRIO_T2": "prio_t2", "REM2D_HEAVY_PER_WAVE": "heavy_per_wave",
                "REM2D_V4_DBG": "debug", "REM2D_REBALANCE_DEV": "rebalance", "REM2D_TRAIN_FAULT": "train_fault"}


def env_options():
    """Experiment overrides for bench.py and tools/: REM2D_PIPELINE, REM2D_FUSE_VELPOST, REM2D_PRIO, REM2D_PRIO_T1 / _T2,
    REM2D_HEAVY_PER_WAVE, REM2D_V4_DBG from the environment as {option: value} -- what BatchedWorld hands to
    rem2d_world_set_option for every world it creates.  The LIBRARY reads no environment variable; this is the only place
    where the variables mean anything."""
    return {opt: int(os.environ[var]) for var, opt in _ENV_OPTIONS.items() if os.environ.get(var, "") != ""}


def env_tile_shape():
    """REM2D_TILE_SHAPE (0 .. 4) as an experiment override of the tile shape, or None."""
    v = os.environ.get("REM2D_TILE_SHAPE", "")
    return int(v) if v != "" else None


def env_tile_creatures():
    """REM2D_TILE_CREATURES: cap on the creatures per velocity tile (rem2d_plan_tiles' max_creatures), 0 = the default."""
    v = os.environ.get("REM2D_TILE_CREATURES", "")
    return int(v) if v != "" else 0

# field ids: order of the enum in include/rem2d.h
FIELDS = [
    "px", "py", "ang", "vx", "vy", "w", "sleept", "hx", "hy", "invm", "invi",
    "fatlx", "fatly", "fatux", "fatuy",
    "jax", "jay", "jbx", "jby", "jtorque", "jlower", "jupper",
    "jimpx", "jimpy", "jimpz", "jmotorimp", "jmotorspeed",
    "shape", "parent", "jround", "awake", "jlimit", "ccount",
    "camp", "cphase", "cfreq", "coffset", "cistate",
    "cedge", "cinfo", "ckey0", "ckey1", "cn0", "cn1", "ct0", "ct1",
    "wod", "fitness", "reward", "done", "everdone", "frozen", "steps", "invdt0",
    "newfix", "err", "positers", "toievents",
]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}

MORPH_FIELDS = ("shape", "hx", "hy", "x", "y", "angle", "parent", "jround", "ax", "ay", "bx", "by", "torque",
                "lower", "upper", "amp", "phase", "freq", "offset", "istate")


class WorldCfg(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("lanes", C.c_int32), ("flags", C.c_uint32), ("device", C.c_int32)]


class Morph(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in MORPH_FIELDS]


class LsystemGenomes(C.Structure):
    """rem2d_lsystem_genomes (include/rem2d.h): host pointers, SoA over genomes."""
    _fields_ = [("n", C.c_int32), ("n_types", C.c_int32)] + [(k, C.c_void_p) for k in (
        "mod_shape", "mod_width", "mod_height", "mod_radius", "mod_angle", "mod_torque", "ctl_amp", "ctl_phase",
        "ctl_freq", "ctl_offset", "rule_n", "rule_site", "rule_ref")]


class TreeBatch(C.Structure):
    """rem2d_tree_batch (include/rem2d.h): host pointers, [n][max_nodes]."""
    _fields_ = [("n", C.c_int32), ("max_nodes", C.c_int32)] + [(k, C.c_void_p) for k in (
        "node_count", "index", "parent", "site", "shape", "width", "height", "radius", "angle", "torque", "ctl_amp",
        "ctl_phase", "ctl_freq", "ctl_offset")]


class NetworkGenomes(C.Structure):
    """rem2d_network_genomes (include/rem2d.h)."""
    _fields_ = ([("n", C.c_int32), ("n_types", C.c_int32), ("n_hidden", C.c_int32), ("max_modules", C.c_int32)]
                + [(k, C.c_void_p) for k in ("w1", "a1", "w2", "mod_shape", "mod_width", "mod_height", "mod_radius",
                                             "mod_angle", "mod_torque", "ctl_amp", "ctl_phase", "ctl_freq", "ctl_offset")]
                + [(k, C.c_double) for k in ("box_min_width", "box_max_width", "box_min_height", "box_max_height",
                                             "box_min_angle", "box_max_angle", "circle_min_radius", "circle_max_radius",
                                             "circle_min_angle", "circle_max_angle", "ctl_max_amp", "ctl_max_phase",
                                             "ctl_max_offset", "ctl_max_freq")])


class TreePopulation(C.Structure):
    """rem2d_tree_population (include/rem2d.h): host pointers, [n][max_nodes], mutated in place by rem2d_mutate_trees."""
    _fields_ = ([("n", C.c_int32), ("max_nodes", C.c_int32)]
                + [(k, C.c_void_p) for k in ("node_count", "parent", "site", "shape", "width", "height", "radius", "angle", "torque",
                                             "ctl_amp", "ctl_phase", "ctl_freq", "ctl_offset")]
                + [(k, C.c_int32) for k in ("max_modules", "max_depth", "n_box", "n_circle")]
                + [(k, C.c_double) for k in ("proto_box_width", "proto_box_height", "proto_circle_radius", "proto_angle",
                                             "proto_torque", "box_min_width", "box_max_width", "box_min_height", "box_max_height",
                                             "box_min_angle", "box_max_angle", "circle_min_radius", "circle_max_radius",
                                             "circle_min_angle", "circle_max_angle", "ctl_max_amp", "ctl_max_phase",
                                             "ctl_max_offset", "ctl_max_freq")])


class StepGroup(C.Structure):
    """rem2d_step_group (include/rem2d.h)."""
    _fields_ = [("worlds", C.POINTER(C.c_void_p)), ("n_worlds", C.c_int32), ("stream", C.c_void_p)]


class Rem2dError(RuntimeError):
    pass


class HandoverError(Rem2dError):
    """A step train launch (REM2D_OPT_FUSE_VELPOST = 2, the default launch form) reported failed hand-overs between the workgroups of
    consecutive steps: a block's previous step was published from another XCD, or a wait for it ran into its 2 s limit (a GPU shared
    with another job, a partition mode that changes the workgroup -> XCD mapping).  NOT a contact-capacity problem: the creatures
    that carry REM2D_ERR_HANDOVER hold a state that cannot be trusted; the remedy is the same creatures on per-step launches
    (``options={"fuse_velpost": 1}``), which evaluate.run_episode does by itself.  ``failures``: workgroups that reported it."""

    def __init__(self, failures, where=""):
        self.failures = int(failures)
        super().__init__("%d hand-over(s) between the workgroups of the step train failed%s: the creatures flagged REM2D_ERR_HANDOVER "
                         "(env.errors() & 4) hold a state that is not to be trusted -- step them on per-step launches instead "
                         "(BatchedModular2D(options={'fuse_velpost': 1}) / REM2D_FUSE_VELPOST=1); evaluate.run_episode re-evaluates "
                         "them that way by itself" % (self.failures, where))


# -fno-slp-vectorize: SLP-packing scalar f32 math into v_pk_* costs more register shuffling (v_mov) than it saves here;
# without it the step kernel fits 238 VGPRs with no spills (+11..14 % env-steps/s).  -ffp-contract=off keeps every binary32
# operation separately rounded (what the bit-exact parity rests on).
# (-instcombine-max-copied-from-constant-users: a kernel reads its by-value `Batch` argument straight from the kernarg segment only
# while LLVM's scan of the argument's users stays under this limit -- 300 by default, which rem2d_step_train_kernel's inlined
# phases exceed: the whole 1.5 KB struct was copied to scratch and spilled from there; the other kernels compile the same either way)
# (-Os, round 6: the same arithmetic -- no fast-math, contraction off -- in ~10 % fewer instructions: +1.3 % on configs 3 / 4 and the
# 8-module chains against -O3; -O2 +0.7 %, -O1 -5 %, -Oz -12 %: profiles/r06_experiments.txt 16)
BUILD_FLAGS = ["-Os", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared",
               "-mllvm", "-instcombine-max-copied-from-constant-users=4000"]
WIDE_FLAGS = ["-DREM2D_WIDE=1"]
FMA_FLAGS = ["-ffp-contract=fast"]   # (the later -ffp-contract wins over BUILD_FLAGS')


def _variant(wide):
    """(path, extra flags, name of the overriding environment variable) of a build: False the default, True the wide-slot
    build, "fma" the tolerance-mode build."""
    if wide == "fma":
        return FMA_LIB_PATH, FMA_FLAGS, "REM2D_FMA_LIB_PATH"
    return (WIDE_LIB_PATH, WIDE_FLAGS, "REM2D_WIDE_LIB_PATH") if wide else (LIB_PATH, [], "REM2D_LIB_PATH")
INCLUDE_DIR = os.path.join(_ROOT, "include")
_ID_MARKER = b"REM2D_BUILD_ID="


def source_id(extra=()):
    """Identity of a build of the library as the sources stand NOW: sha256 over every file under csrc/ (name and bytes, in
    name order), include/rem2d.h and the compile flags, 16 hex digits.  build() compiles it in (-DREM2D_BUILD_ID), lib()
    compares it with rem2d_build_id() and refuses a library built from anything else."""
    import hashlib
    csrc = os.path.dirname(SRC_PATH)
    h = hashlib.sha256()
    for path in sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(INCLUDE_DIR, "rem2d.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(BUILD_FLAGS + list(extra)).encode())
    return h.hexdigest()[:16]


def file_build_id(path):
    """The build id compiled into a library file, read from its bytes (no dlopen: that would pin /opt/rocm's HIP runtime
    before torch has loaded its own); None for a missing file or one without the marker."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(_ID_MARKER)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(_ID_MARKER):j].decode("ascii", "replace")


def build(force=False, verbose=False):
    """Compile csrc/rem2d.hip for gfx950 into gym_rem2d_amd/librem2d.so, (-DREM2D_WIDE) librem2d_wide.so and (-ffp-contract=fast,
    the labelled tolerance mode) librem2d_fma.so -- hipcc cross-compiles without a GPU; the three compile side by side.  A
    library is rebuilt when the id compiled into it differs from source_id(): the hash of the sources and flags, not a file time."""
    procs = []
    for path, extra in ((LIB_PATH, []), (WIDE_LIB_PATH, WIDE_FLAGS), (FMA_LIB_PATH, FMA_FLAGS)):
        want = source_id(extra)
        if not force and file_build_id(path) == want:
            continue
        cmd = ["hipcc"] + BUILD_FLAGS + extra + ['-DREM2D_BUILD_ID="%s"' % want, "-I" + INCLUDE_DIR, SRC_PATH, "-o", path]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    failed = None
    for cmd, pr in procs:   # (wait for every child before raising: no compile is left running behind an exception)
        if pr.wait() != 0 and failed is None:
            failed = (pr.returncode, cmd)
    if failed is not None:
        raise subprocess.CalledProcessError(*failed)
    return LIB_PATH


_lib = None
_libs = {}


def build_id(wide=False):
    """rem2d_build_id() of the loaded library (bench.py prints it)."""
    return lib(wide).rem2d_build_id().decode()


def capacity(wide=False):
    """(REM2D_CONTACT_SLOTS, REM2D_SOLVER_SLOTS) of the default / the wide build."""
    a, b = C.c_int32(), C.c_int32()
    check(lib(wide).rem2d_capacity(C.byref(a), C.byref(b)))
    return a.value, b.value


def lib(wide=False):
    global _lib
    if not wide and _lib is not None:
        return _lib
    if wide and wide in _libs:
        return _libs[wide]
    path, flags, override_var = _variant(wide)
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).
    # Import torch first so that librem2d.so binds to the runtime torch initialises -- two runtimes
    # in one process fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise Rem2dError(
            "gym_rem2d_amd: %s is missing -- the HIP extension is required (there is no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`." % path)
    L = C.CDLL(path)
    L.rem2d_abi_version.restype = C.c_int
    L.rem2d_last_error.restype = C.c_char_p
    L.rem2d_capacity.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rem2d_worlds_launch_info.restype = C.c_int
    L.rem2d_worlds_launch_info.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rem2d_state_bytes.restype = C.c_size_t
    L.rem2d_state_bytes.argtypes = [C.POINTER(WorldCfg)]
    L.rem2d_padded_envs.restype = C.c_int32
    L.rem2d_padded_envs.argtypes = [C.POINTER(WorldCfg)]
    L.rem2d_world_create.argtypes = [C.POINTER(WorldCfg), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.rem2d_world_destroy.argtypes = [C.c_void_p]
    L.rem2d_world_set_terrain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                          C.c_float]
    L.rem2d_world_reset.argtypes = [C.c_void_p, C.POINTER(Morph), C.c_void_p]
    L.rem2d_world_set_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.rem2d_world_set_outputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rem2d_plan_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.rem2d_plan_tiles_shape.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p]
    L.rem2d_world_set_tile_shape.argtypes = [C.c_void_p, C.c_int32]
    L.rem2d_world_set_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.rem2d_world_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.rem2d_world_get_option.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.rem2d_world_adopt.argtypes = [C.c_void_p]
    L.rem2d_world_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.rem2d_world_step_ex.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_void_p]
    L.rem2d_worlds_step.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]
    L.rem2d_worlds_step_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                       C.c_void_p]
    L.rem2d_groups_step.argtypes = [C.POINTER(StepGroup), C.c_int32, C.c_int32, C.c_void_p, C.c_uint32]
    L.rem2d_groups_step_ex.argtypes = [C.POINTER(StepGroup), C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_uint32]
    L.rem2d_compile_lsystem.argtypes = [C.POINTER(LsystemGenomes), C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                        C.POINTER(Morph), C.c_void_p, C.c_int32]
    L.rem2d_compile_trees.argtypes = [C.POINTER(TreeBatch), C.c_double, C.c_int32, C.POINTER(Morph), C.c_void_p, C.c_int32]
    L.rem2d_compile_network.argtypes = [C.POINTER(NetworkGenomes), C.c_int32, C.c_double, C.c_int32, C.POINTER(Morph),
                                        C.c_void_p, C.c_int32]
    L.rem2d_mutate_trees.argtypes = [C.POINTER(TreePopulation), C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_int32]
    L.rem2d_tree_diversity.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.rem2d_selftest_scalar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.rem2d_world_field.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_int32)]
    L.rem2d_world_enable_timing.argtypes = [C.c_void_p, C.c_int32]
    L.rem2d_world_kernel_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.rem2d_world_step_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.rem2d_world_handover_failures.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    if L.rem2d_abi_version() != 11:
        raise Rem2dError("%s: ABI version mismatch" % os.path.basename(path))
    L.rem2d_build_id.restype = C.c_char_p
    # the library must have been built from the sources beside it (REM2D_LIB_PATH / REM2D_WIDE_LIB_PATH name an experiment's
    # variant build on purpose: tools/build_variant.sh, exempt)
    overridden = os.environ.get(override_var)
    have, want = L.rem2d_build_id().decode(), source_id(flags)
    if not overridden and have != want:
        raise Rem2dError("%s is stale: built from sources with id %s, the sources here have id %s -- rebuild it with "
                         "`python -c 'import __graft_entry__ as g; g.build()'`" % (os.path.basename(path), have, want))
    if wide:
        _libs[wide] = L
    else:
        _lib = L
    return L


def plan_tiles(parent, jround, n_envs, lanes, n_padded, max_creatures=0, tile_shape=-1, wide=False):
    """Tile plan of the velocity kernel for one morphology batch (rem2d_plan_tiles_shape): int32 tile starts
    [n_tiles + 1].  tile_shape: 0 .. 4, or -1 for the default (3); max_creatures 0: the library's default cap unless
    REM2D_TILE_CREATURES overrides it (experiments)."""
    import numpy as np
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    jround = np.ascontiguousarray(jround, dtype=np.int32)
    out = np.zeros(int(n_padded) + 1, dtype=np.int32)
    n = C.c_int32()
    if max_creatures <= 0:
        max_creatures = env_tile_creatures()
    # (the plan of the build that will run it: the wide build schedules one phase more than the default one, V4_PHASES)
    check(lib(wide).rem2d_plan_tiles_shape(parent.ctypes.data, jround.ctypes.data, int(n_envs), int(lanes), int(n_padded),
                                           int(max_creatures), int(tile_shape), out.ctypes.data, C.byref(n)), wide)
    return out[:n.value + 1].copy()


def check(rc, wide=False):
    if rc != 0:
        raise Rem2dError("librem2d: error %d: %s" % (rc, lib(wide).rem2d_last_error().decode()))
