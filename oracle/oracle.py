"""ctypes binding of oracle/librem2d_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/rem2d_oracle.h).  The product package gym_rem2d_amd
never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librem2d_oracle.so")
_SO_F64 = os.path.join(_HERE, "librem2d_oracle_f64.so")   # binary64 "truth" build of the same source

FLAG_CONTINUOUS = 1
FLAG_SLEEP_RESET_ALWAYS = 2
FLAG_NO_SLEEP = 4
FLAG_TOI_TRANSPARENT_STATICS = 8

MORPH_F32 = ("hx", "hy", "x", "y", "angle", "ax", "ay", "bx", "by", "torque", "lower", "upper")
MORPH_F64 = ("amp", "phase", "freq", "offset", "istate")


class OMorph(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("lanes", C.c_int32), ("shape", C.c_void_p),
                ("hx", C.c_void_p), ("hy", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p),
                ("angle", C.c_void_p), ("parent", C.c_void_p),
                ("ax", C.c_void_p), ("ay", C.c_void_p), ("bx", C.c_void_p), ("by", C.c_void_p),
                ("torque", C.c_void_p), ("lower", C.c_void_p), ("upper", C.c_void_p),
                ("amp", C.c_void_p), ("phase", C.c_void_p), ("freq", C.c_void_p),
                ("offset", C.c_void_p), ("istate", C.c_void_p)]


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = [os.path.join(_HERE, "rem2d_oracle.c"), os.path.join(_HERE, "rem2d_oracle.h")]
    newest = max(os.path.getmtime(f) for f in src)
    for so in (_SO, _SO_F64):
        if force or not os.path.exists(so) or os.path.getmtime(so) < newest:
            subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(so)], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_lib_f64 = None


def lib_f64():
    """The binary64 build (every engine quantity a double; same ABI).  Not bit-comparable with anything: it measures
    how much of a trajectory difference is inherent to binary32 (SURVEY.md 8c protocol iv)."""
    global _lib_f64
    if _lib_f64 is None:
        if not os.path.exists(_SO_F64):
            build()
        _lib_f64 = _load(_SO_F64)
        assert _lib_f64.rem2d_oracle_is_f64() == 1
    return _lib_f64


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = _load(_SO)
    return _lib


def _load(path):
    if True:   # (indentation kept: the block below configures one CDLL instance)
        L = C.CDLL(path)
        L.rem2d_oracle_terrain_create.restype = C.c_void_p
        L.rem2d_oracle_terrain_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float]
        L.rem2d_oracle_terrain_destroy.argtypes = [C.c_void_p]
        L.rem2d_oracle_world_create.restype = C.c_void_p
        L.rem2d_oracle_world_create.argtypes = [C.c_void_p, C.c_uint]
        L.rem2d_oracle_world_destroy.argtypes = [C.c_void_p]
        L.rem2d_oracle_add_box.argtypes = [C.c_void_p] + [C.c_float] * 5
        L.rem2d_oracle_add_circle.argtypes = [C.c_void_p] + [C.c_float] * 4
        L.rem2d_oracle_add_joint.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 7
        L.rem2d_oracle_set_controller.argtypes = [C.c_void_p, C.c_int] + [C.c_double] * 5
        L.rem2d_oracle_set_motor_speed.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.rem2d_oracle_set_velocity.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 3
        L.rem2d_oracle_set_gravity.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.rem2d_oracle_set_body_state.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 6 + [C.c_int]
        L.rem2d_oracle_set_joint_impulses.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4
        L.rem2d_oracle_set_contact_impulses.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 4
        L.rem2d_oracle_get_wod.restype = C.c_double
        L.rem2d_oracle_get_wod.argtypes = [C.c_void_p]
        L.rem2d_oracle_get_controller_state.argtypes = [C.c_void_p, C.c_void_p]
        L.rem2d_oracle_world_step.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
        L.rem2d_oracle_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rem2d_oracle_env_step_ex.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rem2d_oracle_batch_toi_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for f in ("num_bodies", "num_joints", "position_iterations", "toi_events", "toi_dynamic_advances"):
            getattr(L, "rem2d_oracle_" + f).argtypes = [C.c_void_p]
        for f in ("get_bodies", "get_mass", "get_joints"):
            getattr(L, "rem2d_oracle_" + f).argtypes = [C.c_void_p, C.c_void_p]
        L.rem2d_oracle_get_island_joint_order.argtypes = [C.c_void_p, C.c_void_p]
        L.rem2d_oracle_get_contacts.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rem2d_oracle_get_manifold.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rem2d_oracle_get_fat_aabb.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rem2d_oracle_sincosf.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        L.rem2d_oracle_sin.restype = C.c_double
        L.rem2d_oracle_sin.argtypes = [C.c_double]
        L.rem2d_oracle_box_mass.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.rem2d_oracle_circle_mass.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        L.rem2d_oracle_is_f64.restype = C.c_int
        for f in ("kat_distance", "kat_collide"):
            getattr(L, "rem2d_oracle_" + f).argtypes = [C.c_void_p] * (5 if f == "kat_distance" else 6)
        L.rem2d_oracle_kat_toi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.rem2d_oracle_kat_contact_solve.argtypes = [C.c_void_p, C.c_void_p]
        L.rem2d_oracle_kat_scalar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.rem2d_oracle_batch_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint] + [C.c_void_p] * 5
        L.rem2d_oracle_batch_run_caps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint] + [C.c_void_p] * 5
        L.rem2d_oracle_batch_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_void_p,
                                                C.c_void_p]
        L.rem2d_oracle_world_from_morph.restype = C.c_void_p
        L.rem2d_oracle_world_from_morph.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Terrain:
    """Static terrain: npts heights -> npts-1 edge bodies (+ hardcore boxes)."""

    def __init__(self, xs, ys, polys=None, friction=2.5, f64=False):
        self.xs = np.ascontiguousarray(xs, dtype=np.float32)
        self.ys = np.ascontiguousarray(ys, dtype=np.float32)
        self.polys = np.ascontiguousarray(polys if polys is not None else np.zeros((0, 4, 2)), dtype=np.float32)
        self.L = lib_f64() if f64 else lib()   # a terrain belongs to the build that created it
        self.h = self.L.rem2d_oracle_terrain_create(_ptr(self.xs), _ptr(self.ys), len(self.xs),
                                                    _ptr(self.polys), len(self.polys), friction)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.rem2d_oracle_terrain_destroy(self.h)
            self.h = None


def make_omorph(m):
    """m: dict of numpy arrays in [env][lane] layout (see gym_rem2d_amd.compiler.Morphology.arrays)."""
    keep = {}
    om = OMorph()
    om.n_envs = int(m["n_envs"])
    om.lanes = int(m["lanes"])
    for k in ("shape", "parent"):
        keep[k] = np.ascontiguousarray(m[k], dtype=np.int32)
        setattr(om, k, keep[k].ctypes.data)
    for k in MORPH_F32:
        keep[k] = np.ascontiguousarray(m[k], dtype=np.float32)
        setattr(om, k, keep[k].ctypes.data)
    for k in MORPH_F64:
        keep[k] = np.ascontiguousarray(m[k], dtype=np.float64)
        setattr(om, k, keep[k].ctypes.data)
    return om, keep


class World:
    def __init__(self, terrain, flags=0, handle=None):
        self.terrain = terrain
        self.h = handle if handle is not None else lib().rem2d_oracle_world_create(terrain.h, flags)

    @classmethod
    def from_morph(cls, terrain, m, env, flags=0):
        om, keep = make_omorph(m)
        h = lib().rem2d_oracle_world_from_morph(terrain.h, C.byref(om), env, flags)
        return cls(terrain, flags, handle=h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().rem2d_oracle_world_destroy(self.h)
            self.h = None

    def add_box(self, hx, hy, x, y, angle=0.0):
        return lib().rem2d_oracle_add_box(self.h, hx, hy, x, y, angle)

    def add_circle(self, r, x, y, angle=0.0):
        return lib().rem2d_oracle_add_circle(self.h, r, x, y, angle)

    def add_joint(self, a, b, ax, ay, bx, by, torque=50.0, lower=-np.pi / 2, upper=np.pi / 2):
        return lib().rem2d_oracle_add_joint(self.h, a, b, ax, ay, bx, by, torque, lower, upper)

    def set_controller(self, joint, amp, phase, freq, offset, istate=0.0):
        lib().rem2d_oracle_set_controller(self.h, joint, amp, phase, freq, offset, istate)

    def set_motor_speed(self, joint, speed):
        lib().rem2d_oracle_set_motor_speed(self.h, joint, speed)

    def set_velocity(self, body, vx, vy, w):
        lib().rem2d_oracle_set_velocity(self.h, body, vx, vy, w)

    def set_gravity(self, gx, gy):
        lib().rem2d_oracle_set_gravity(self.h, gx, gy)

    # ---- state re-synchronisation (tests/test_box2d_pin.py): exactly what pybox2d lets a script read from a b2World ----
    def set_body_state(self, body, x, y, angle, vx, vy, w, awake=1):
        lib().rem2d_oracle_set_body_state(self.h, body, x, y, angle, vx, vy, w, int(awake))

    def set_joint_impulses(self, joint, ix, iy, iz, motor_impulse):
        lib().rem2d_oracle_set_joint_impulses(self.h, joint, ix, iy, iz, motor_impulse)

    def set_contact_impulses(self, body, k, n0, n1, t0, t1):
        lib().rem2d_oracle_set_contact_impulses(self.h, body, k, n0, n1, t0, t1)

    @property
    def wod(self):
        return lib().rem2d_oracle_get_wod(self.h)

    def controller_state(self):
        out = np.zeros(max(1, self.n_joints), dtype=np.float64)
        lib().rem2d_oracle_get_controller_state(self.h, _ptr(out))
        return out[:self.n_joints].copy()

    def step(self, dt=1.0 / 50, vel_iters=180, pos_iters=60):
        lib().rem2d_oracle_world_step(self.h, dt, vel_iters, pos_iters)

    def env_step_ex(self, dt, vel_iters, pos_iters):
        r, d = C.c_double(), C.c_int()
        lib().rem2d_oracle_env_step_ex(self.h, dt, vel_iters, pos_iters, C.byref(r), C.byref(d))
        return r.value, d.value

    def env_step(self):
        r = C.c_double()
        d = C.c_int()
        lib().rem2d_oracle_env_step(self.h, C.byref(r), C.byref(d))
        return r.value, d.value

    @property
    def n_bodies(self):
        return lib().rem2d_oracle_num_bodies(self.h)

    @property
    def n_joints(self):
        return lib().rem2d_oracle_num_joints(self.h)

    def bodies(self):
        out = np.zeros((self.n_bodies, 8), dtype=np.float32)
        lib().rem2d_oracle_get_bodies(self.h, _ptr(out))
        return out

    def mass(self):
        out = np.zeros((self.n_bodies, 4), dtype=np.float32)
        lib().rem2d_oracle_get_mass(self.h, _ptr(out))
        return out

    def joints(self):
        out = np.zeros((self.n_joints, 6), dtype=np.float32)
        lib().rem2d_oracle_get_joints(self.h, _ptr(out))
        return out

    def island_joint_order(self):
        out = np.zeros(128, dtype=np.int32)
        n = lib().rem2d_oracle_get_island_joint_order(self.h, _ptr(out))
        return out[:n].copy()

    def contacts(self, body):
        out = np.zeros((24, 8), dtype=np.int32)
        fout = np.zeros((24, 4), dtype=np.float32)
        n = lib().rem2d_oracle_get_contacts(self.h, body, _ptr(out), _ptr(fout))
        return out[:n].copy(), fout[:n].copy()

    def manifold(self, body, k):
        out = np.zeros(8, dtype=np.float32)
        lib().rem2d_oracle_get_manifold(self.h, body, k, _ptr(out))
        return out

    def fat_aabb(self, body):
        out = np.zeros(4, dtype=np.float32)
        lib().rem2d_oracle_get_fat_aabb(self.h, body, _ptr(out))
        return out

    @property
    def position_iterations(self):
        return lib().rem2d_oracle_position_iterations(self.h)

    @property
    def toi_events(self):
        return lib().rem2d_oracle_toi_events(self.h)

    @property
    def toi_dynamic_advances(self):
        return lib().rem2d_oracle_toi_dynamic_advances(self.h)


# ---- standalone pieces for the known-answer tests ----
TOI_STATES = ("unknown", "failed", "overlapped", "touching", "separated")


def _spec(shape):
    """("edge", x1, y1, x2, y2) | ("box", hx, hy) | ("circle", r) | ("poly", [(x, y), ...]) -> float32 spec."""
    kind = shape[0]
    if kind == "edge":
        v = [0.0] + list(shape[1:5])
    elif kind == "box":
        v = [1.0, shape[1], shape[2]]
    elif kind == "circle":
        v = [2.0, shape[1]]
    else:
        pts = shape[1]
        v = [3.0, float(len(pts))] + [c for p in pts for c in p]
    return np.asarray(v, dtype=np.float32)


def distance(shapeA, xfA, shapeB, xfB):
    """b2Distance (useRadii=False) -> (pointA, pointB, distance, iterations); xf = (x, y, angle)."""
    out = np.zeros(6, dtype=np.float32)
    a, b = _spec(shapeA), _spec(shapeB)
    xa, xb = np.asarray(xfA, dtype=np.float32), np.asarray(xfB, dtype=np.float32)
    assert lib().rem2d_oracle_kat_distance(_ptr(a), _ptr(xa), _ptr(b), _ptr(xb), _ptr(out)) == 0
    return out[0:2].copy(), out[2:4].copy(), float(out[4]), int(out[5])


def time_of_impact(shapeA, sweepA, shapeB, sweepB, t_max=1.0):
    """b2TimeOfImpact -> (state name, t); sweep = (c0x, c0y, a0, cx, cy, a)."""
    out = np.zeros(2, dtype=np.float32)
    a, b = _spec(shapeA), _spec(shapeB)
    sa, sb = np.asarray(sweepA, dtype=np.float32), np.asarray(sweepB, dtype=np.float32)
    assert lib().rem2d_oracle_kat_toi(_ptr(a), _ptr(sa), _ptr(b), _ptr(sb), t_max, _ptr(out)) == 0
    return TOI_STATES[int(out[0])], float(out[1])


def collide(shapeA, xfA, shapeB, xfB):
    """narrowphase -> dict(type, count, keys, normal, point, points) in manifold-local coordinates."""
    io, fo = np.zeros(4, dtype=np.int32), np.zeros(8, dtype=np.float32)
    a, b = _spec(shapeA), _spec(shapeB)
    xa, xb = np.asarray(xfA, dtype=np.float32), np.asarray(xfB, dtype=np.float32)
    assert lib().rem2d_oracle_kat_collide(_ptr(a), _ptr(xa), _ptr(b), _ptr(xb), _ptr(io), _ptr(fo)) == 0
    return dict(type=int(io[0]), count=int(io[1]), keys=(int(io[2]) & 0xffffffff, int(io[3]) & 0xffffffff),
                normal=fo[0:2].copy(), point=fo[2:4].copy(), points=fo[4:8].reshape(2, 2).copy())


def contact_solve(normal, points, cB, inv_mass, inv_I, friction, vB, wB, n_imp=(0.0, 0.0), t_imp=(0.0, 0.0)):
    """One SolveVelocityConstraints sweep of one static-vs-body contact -> (v, w, normalImpulses, tangentImpulses, count)."""
    pts = list(points) + [(0.0, 0.0)] * (2 - len(points))
    vin = np.asarray([normal[0], normal[1], len(points), pts[0][0], pts[0][1], pts[1][0], pts[1][1], cB[0], cB[1],
                      inv_mass, inv_I, friction, vB[0], vB[1], wB, n_imp[0], n_imp[1], t_imp[0], t_imp[1]], dtype=np.float32)
    out = np.zeros(8, dtype=np.float32)
    assert lib().rem2d_oracle_kat_contact_solve(_ptr(vin), _ptr(out)) == 0
    return out[0:2].copy(), float(out[2]), out[3:5].copy(), out[5:7].copy(), int(out[7])


def batch_toi_stats(reset=True):
    """(TOI sub-steps, forced dynamic-sweep advances) over all batch_run worlds since the last reset."""
    ev, adv = C.c_longlong(), C.c_longlong()
    lib().rem2d_oracle_batch_toi_stats(C.byref(ev), C.byref(adv), 1 if reset else 0)
    return ev.value, adv.value


def kat_scalar(a, b, c):
    """(b2Min(a, b), b2Max(a, b), b2Clamp(a, b, c)) element-wise in the oracle's own form: float32 [3, n]."""
    a, b, c = (np.ascontiguousarray(v, dtype=np.float32) for v in (a, b, c))
    out = np.zeros((3, len(a)), dtype=np.float32)
    assert lib().rem2d_oracle_kat_scalar(_ptr(a), _ptr(b), _ptr(c), len(a), _ptr(out)) == 0
    return out


def sincosf(a):
    s = C.c_float()
    c = C.c_float()
    lib().rem2d_oracle_sincosf(a, C.byref(s), C.byref(c))
    return s.value, c.value


def sin64(x):
    return lib().rem2d_oracle_sin(x)


def box_mass(hx, hy):
    m = C.c_float()
    i = C.c_float()
    lib().rem2d_oracle_box_mass(hx, hy, C.byref(m), C.byref(i))
    return m.value, i.value


def circle_mass(r):
    m = C.c_float()
    i = C.c_float()
    lib().rem2d_oracle_circle_mass(r, C.byref(m), C.byref(i))
    return m.value, i.value


def batch_run(terrain, m, n_steps, n_threads=1, flags=0, trace=False):
    """Run Modular2D.step n_steps times for every env of morphology dict m.

    Returns dict(bodies [N,K,8], reward [N], done [N], fitness [N], trace [T,N,K,3] or None)."""
    om, keep = make_omorph(m)
    N, K = om.n_envs, om.lanes
    bodies = np.zeros((N, K, 8), dtype=np.float32)
    reward = np.zeros(N, dtype=np.float64)
    done = np.zeros(N, dtype=np.int32)
    fitness = np.zeros(N, dtype=np.float64)
    tr = np.zeros((n_steps, N, K, 3), dtype=np.float32) if trace else None
    rc = terrain.L.rem2d_oracle_batch_run(terrain.h, C.byref(om), n_steps, n_threads, flags, _ptr(bodies),
                                      _ptr(reward), _ptr(done), _ptr(fitness),
                                      _ptr(tr) if trace else None)
    if rc != 0:
        raise RuntimeError("rem2d_oracle_batch_run failed: %d" % rc)
    return dict(bodies=bodies, reward=reward, done=done, fitness=fitness, trace=tr)


def batch_run_caps(terrain, m, n_steps, n_threads=1, flags=0):
    """batch_run plus ``caps`` [N, 3]: the most pairs / touching manifolds any body of the creature held while its fitness
    was still open, and the pairs the oracle itself refused (> O_MAX_BODY_CONTACTS = 32).  Bookkeeping, not arithmetic."""
    om, keep = make_omorph(m)
    N, K = om.n_envs, om.lanes
    bodies = np.zeros((N, K, 8), dtype=np.float32)
    reward = np.zeros(N, dtype=np.float64)
    done = np.zeros(N, dtype=np.int32)
    fitness = np.zeros(N, dtype=np.float64)
    caps = np.zeros((N, 3), dtype=np.int32)
    rc = terrain.L.rem2d_oracle_batch_run_caps(terrain.h, C.byref(om), n_steps, n_threads, flags, _ptr(bodies),
                                           _ptr(reward), _ptr(done), _ptr(fitness), _ptr(caps))
    if rc != 0:
        raise RuntimeError("rem2d_oracle_batch_run_caps failed: %d" % rc)
    return dict(bodies=bodies, reward=reward, done=done, fitness=fitness, caps=caps)


def batch_window(terrain, m, settle, window, n_threads=1, flags=0):
    """bench.py's cpu_baseline: `settle` untimed steps, then ONE continuous timed window of `window` steps of every
    env of morphology dict m.  Returns (seconds of the window, reward [N] after it)."""
    om, keep = make_omorph(m)
    sec = C.c_double(0.0)
    reward = np.zeros(om.n_envs, dtype=np.float64)
    rc = terrain.L.rem2d_oracle_batch_window(terrain.h, C.byref(om), int(settle), int(window), int(n_threads), flags,
                                             C.byref(sec), _ptr(reward))
    if rc != 0:
        raise RuntimeError("rem2d_oracle_batch_window failed: %d" % rc)
    return sec.value, reward
