"""ctypes wrapper of oracle/librem2d_cpu.so: include/rem2d.h's ABI on host pointers, backed by the oracle.

TEST INFRASTRUCTURE ONLY (see rem2d_cpu.c).  ``CpuWorld`` mirrors ``gym_rem2d_amd.world.BatchedWorld`` call for
call -- same constructor arguments, ``set_terrain / reset / step / step_ex / view`` -- with a numpy arena instead of
a device tensor, so that one test body can drive the HIP library and the twin and compare their arenas."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librem2d_cpu.so")

# the same tables as gym_rem2d_amd/_lib.py (order of the enums in include/rem2d.h); restated here because nothing under
# oracle/ imports the product package and vice versa
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
CONTACT_SLOTS = 24
_DTYPES = {0: np.float32, 1: np.int32, 2: np.float64}
# what the twin does not maintain (rem2d_cpu.c header): compared by no test
UNMAINTAINED = ("newfix", "err")


class WorldCfg(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("lanes", C.c_int32), ("flags", C.c_uint32), ("device", C.c_int32)]


class Morph(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in MORPH_FIELDS]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("rem2d_cpu.c", "rem2d_oracle.c", "rem2d_oracle.h")]
    src.append(os.path.join(_HERE, "..", "include", "rem2d.h"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_SO)], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.rem2d_cpu_last_error.restype = C.c_char_p
        L.rem2d_cpu_state_bytes.restype = C.c_size_t
        L.rem2d_cpu_state_bytes.argtypes = [C.POINTER(WorldCfg)]
        L.rem2d_cpu_padded_envs.restype = C.c_int32
        L.rem2d_cpu_padded_envs.argtypes = [C.POINTER(WorldCfg)]
        L.rem2d_cpu_world_create.argtypes = [C.POINTER(WorldCfg), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.rem2d_cpu_world_destroy.argtypes = [C.c_void_p]
        L.rem2d_cpu_world_set_terrain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                  C.c_float]
        L.rem2d_cpu_world_reset.argtypes = [C.c_void_p, C.POINTER(Morph), C.c_void_p]
        L.rem2d_cpu_world_set_outputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rem2d_cpu_world_set_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.rem2d_cpu_world_set_tile_shape.argtypes = [C.c_void_p, C.c_int32]
        L.rem2d_cpu_world_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.rem2d_cpu_world_step_ex.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_void_p]
        L.rem2d_cpu_worlds_step.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]
        L.rem2d_cpu_world_field.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_int32)]
        _lib = L
    return _lib


class CpuTwinError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise CpuTwinError("rem2d_cpu error %d: %s" % (rc, lib().rem2d_cpu_last_error().decode()))


class CpuWorld:
    def __init__(self, n_envs, lanes, flags=0):
        self.n_envs, self.lanes, self.flags = int(n_envs), int(lanes), int(flags)
        L = lib()
        self.cfg = WorldCfg(self.n_envs, self.lanes, self.flags, 0)
        nbytes = L.rem2d_cpu_state_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise CpuTwinError("invalid world shape: n_envs=%d lanes=%d" % (n_envs, lanes))
        self.n_envs_padded = L.rem2d_cpu_padded_envs(C.byref(self.cfg))
        self.arena = np.zeros(nbytes, dtype=np.uint8)
        h = C.c_void_p()
        check(L.rem2d_cpu_world_create(C.byref(self.cfg), self.arena.ctypes.data, nbytes, C.byref(h)))
        self.h = h
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            lib().rem2d_cpu_world_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_terrain(self, terrain):
        xs, ys, polys = terrain.f32()
        xs, ys, polys = np.ascontiguousarray(xs), np.ascontiguousarray(ys), np.ascontiguousarray(polys)
        check(lib().rem2d_cpu_world_set_terrain(self.h, xs.ctypes.data, ys.ctypes.data, len(xs),
                                                polys.ctypes.data if len(polys) else None, len(polys),
                                                float(terrain.friction)))

    def reset(self, morph):
        """morph: gym_rem2d_amd.compiler.Morphology (its ``arrays`` dict of host numpy arrays)."""
        m = Morph()
        keep = {k: np.ascontiguousarray(morph.arrays[k]) for k in MORPH_FIELDS}
        for k in MORPH_FIELDS:
            setattr(m, k, keep[k].ctypes.data)
        self._keep = keep
        check(lib().rem2d_cpu_world_reset(self.h, C.byref(m), None))

    def set_outputs(self, reward, done, index):
        self._outputs = (reward, done, index)
        check(lib().rem2d_cpu_world_set_outputs(self.h, reward.ctypes.data, done.ctypes.data, index.ctypes.data))

    def step(self, n_steps=1):
        check(lib().rem2d_cpu_world_step(self.h, int(n_steps), None))

    def step_ex(self, n_steps, dt, vel_iters, pos_iters):
        check(lib().rem2d_cpu_world_step_ex(self.h, int(n_steps), float(dt), int(vel_iters), int(pos_iters), None))

    def field(self, name):
        off, cnt, dt = C.c_size_t(), C.c_size_t(), C.c_int32()
        check(lib().rem2d_cpu_world_field(self.h, FIELD_ID[name], C.byref(off), C.byref(cnt), C.byref(dt)))
        return off.value, cnt.value, dt.value

    def view(self, name):
        off, cnt, dt = self.field(name)
        nb = cnt * (8 if dt == 2 else 4)
        v = self.arena[off:off + nb].view(_DTYPES[dt])
        Lp = self.n_envs_padded * self.lanes
        if cnt == Lp:
            return v.reshape(self.n_envs_padded, self.lanes)[:self.n_envs]
        if cnt == Lp * CONTACT_SLOTS:
            return v.reshape(CONTACT_SLOTS, self.n_envs_padded, self.lanes)[:, :self.n_envs]
        return v[:self.n_envs]

    def bodies(self):
        cols = [self.view(k).astype(np.float32) for k in ("px", "py", "ang", "vx", "vy", "w", "sleept")]
        cols.append(self.view("awake").astype(np.float32))
        return np.stack(cols, axis=-1)
