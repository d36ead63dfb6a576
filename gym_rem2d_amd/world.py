"""BatchedWorld: N independent Box2D-style worlds stepped in lockstep on one MI355X.

Thin host wrapper over the C ABI (include/rem2d.h).  Device memory is a single torch.uint8
arena owned by this object; state fields are exposed as zero-copy torch views
(``world.view("px")``).  PyTorch is used for memory, streams and (elsewhere)
torch.distributed only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .compiler import MORPH_F32, MORPH_F64, MORPH_I32, Morphology
from .terrain import TerrainProfile

_DTYPES = {0: torch.float32, 1: torch.int32, 2: torch.float64}


class BatchedWorld:
    def __init__(self, n_envs, lanes, flags=0, device=None, wide=False, options=None):
        """wide: True = librem2d_wide.so (32 pair slots / 12 solver slots per body instead of 24 / 6); "fma" = librem2d_fma.so, the
        -ffp-contract=fast tolerance mode (NOT bit-exact: validated within a stated tolerance, never the default).  options: launch options
        {name: value} (``_lib.OPTIONS``, rem2d_world_set_option) -- launch shapes and scheduling hints, never results;
        on top of the REM2D_* experiment overrides of the environment (``_lib.env_options``)."""
        if not torch.cuda.is_available():
            raise _lib.Rem2dError("gym_rem2d_amd needs a ROCm GPU (MI355X); no CPU fallback exists")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.n_envs, self.lanes, self.flags = int(n_envs), int(lanes), int(flags)
        self.wide = wide if wide == "fma" else bool(wide)   # ("fma": the labelled tolerance-mode build, _lib.FMA_LIB_PATH)
        L = self.L = _lib.lib(self.wide)
        self.contact_slots = _lib.capacity(self.wide)[0]
        self.cfg = _lib.WorldCfg(self.n_envs, self.lanes, self.flags, self.device.index or 0)
        nbytes = L.rem2d_state_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise _lib.Rem2dError("invalid world shape: n_envs=%d lanes=%d" % (n_envs, lanes))
        self.n_envs_padded = L.rem2d_padded_envs(C.byref(self.cfg))
        self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        self._check(L.rem2d_world_create(C.byref(self.cfg), self.arena.data_ptr(), nbytes, C.byref(h)))
        self.h = h
        self._views = {}
        self._morph_dev = None
        self.terrain = None
        shape = _lib.env_tile_shape()   # experiment override; None: the library's default (64-lane tiles)
        self.tile_shape = -1
        if shape is not None:
            self._check(L.rem2d_world_set_tile_shape(self.h, shape))
            self.tile_shape = shape
        for name, value in dict(_lib.env_options(), **(options or {})).items():
            self.set_option(name, value)

    def set_order(self, order, check=True):
        """Install a creature order (rem2d_world_set_order): slot e of the velocity tiles / position blocks handles creature
        order[e].  order: int tensor [n_envs] -- a permutation of the creatures (verified unless check=False) -- on any
        device, or None for the identity.  Padding slots keep themselves.  A launch shape: results do not depend on it."""
        if order is None:
            self._order = None
            self._check(self.L.rem2d_world_set_order(self.h, None, self._stream()))
            return
        order = order.to(device=self.device, dtype=torch.int64)
        if check:   # the library trusts the array (a duplicate would make two slots step one creature): verify here
            if order.numel() != self.n_envs or not bool(torch.equal(torch.sort(order).values,
                                                                       torch.arange(self.n_envs, device=self.device))):
                raise ValueError("set_order: `order` must be a permutation of range(n_envs)")
        full = torch.arange(self.n_envs_padded, dtype=torch.int32, device=self.device)
        full[:self.n_envs] = order.to(torch.int32)
        self._order = full   # (kept alive until the asynchronous copy has run)
        self._check(self.L.rem2d_world_set_order(self.h, full.data_ptr(), self._stream()))

    def set_option(self, name, value):
        """A launch option of this world (``_lib.OPTIONS``: pipeline, fuse_velpost, prio, prio_t1, prio_t2, heavy_per_wave,
        debug); in a step group the first world's options steer the group's launches."""
        self._check(self.L.rem2d_world_set_option(self.h, _lib.OPTION_ID[name], int(value)))

    def get_option(self, name):
        v = C.c_int32()
        self._check(self.L.rem2d_world_get_option(self.h, _lib.OPTION_ID[name], C.byref(v)))
        return v.value

    def _check(self, rc):
        _lib.check(rc, self.wide)

    def close(self):
        if getattr(self, "h", None):
            self.L.rem2d_world_destroy(self.h)
            self.h = None

    def release(self):
        """close() and drop the device memory (arena, cached views, uploaded morphology) right away."""
        self.close()
        self.arena = None
        self._views.clear()
        self._morph_dev = None
        self._outputs = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- terrain: _generate_terrain's static bodies, shared by every world ----
    def set_terrain(self, terrain: TerrainProfile):
        xs, ys, polys = terrain.f32()
        xs, ys, polys = np.ascontiguousarray(xs), np.ascontiguousarray(ys), np.ascontiguousarray(polys)
        self._check(self.L.rem2d_world_set_terrain(
            self.h, xs.ctypes.data, ys.ctypes.data, len(xs), polys.ctypes.data if len(polys) else None, len(polys),
            float(terrain.friction)))
        self.terrain = terrain

    # ---- reset: upload the morphology and rebuild every world ----
    def reset(self, morph: Morphology, tile_shape=None):
        """tile_shape: launch shape of the velocity kernel for this world (0 .. 4, include/rem2d.h
        rem2d_world_set_tile_shape); None keeps the world's current one (3 unless REM2D_TILE_SHAPE overrode it)."""
        if morph.n_envs != self.n_envs or morph.lanes != self.lanes:
            raise ValueError("morphology shape (%d x %d) does not match world (%d x %d)" %
                             (morph.n_envs, morph.lanes, self.n_envs, self.lanes))
        dev = {}
        for k in MORPH_I32 + MORPH_F32 + MORPH_F64:
            dev[k] = torch.from_numpy(morph.arrays[k]).to(self.device, non_blocking=False)
        m = _lib.Morph()
        for k in _lib.MORPH_FIELDS:
            setattr(m, k, dev[k].data_ptr())
        self._morph_dev = dev  # keep alive until the reset kernel has run
        self._check(self.L.rem2d_world_reset(self.h, C.byref(m), self._stream()))
        # work partition of the velocity kernel: consecutive creatures packed into tiles of <= 256 lanes with <= 64
        # joints per schedule phase (include/rem2d.h, rem2d_world_set_tiles)
        if tile_shape is not None:
            self._check(self.L.rem2d_world_set_tile_shape(self.h, int(tile_shape)))
            self.tile_shape = int(tile_shape)
        self.tiles = _lib.plan_tiles(morph.arrays["parent"], morph.arrays["jround"], self.n_envs, self.lanes,
                                     self.n_envs_padded, tile_shape=self.tile_shape, wide=self.wide)
        self._check(self.L.rem2d_world_set_tiles(self.h, self.tiles.ctypes.data, len(self.tiles) - 1))

    def adopt(self, morph: Morphology, tile_shape=None):
        """Instead of reset(): the caller has filled every state field of this world (``view(name)`` for all of
        ``_lib.FIELDS``) from another world between two steps (include/rem2d.h, rem2d_world_adopt).  ``morph``: the
        host-side layout of the creatures now in this world, for the tile plan of the velocity kernel."""
        if morph.n_envs != self.n_envs or morph.lanes != self.lanes:
            raise ValueError("morphology shape (%d x %d) does not match world (%d x %d)" %
                             (morph.n_envs, morph.lanes, self.n_envs, self.lanes))
        self._check(self.L.rem2d_world_adopt(self.h))
        if tile_shape is not None:
            self._check(self.L.rem2d_world_set_tile_shape(self.h, int(tile_shape)))
            self.tile_shape = int(tile_shape)
        self.tiles = _lib.plan_tiles(morph.arrays["parent"], morph.arrays["jround"], self.n_envs, self.lanes,
                                     self.n_envs_padded, tile_shape=self.tile_shape, wide=self.wide)
        self._check(self.L.rem2d_world_set_tiles(self.h, self.tiles.ctypes.data, len(self.tiles) - 1))

    def set_outputs(self, reward, done, index):
        """Let the kernels also write reward / done of creature e to reward[index[e]] / done[index[e]] (population
        order; `done` is a torch.bool tensor, `index` int32 on the device).  The tensors are kept alive here."""
        self._outputs = (reward, done, index)
        self._check(self.L.rem2d_world_set_outputs(self.h, reward.data_ptr(), done.data_ptr(), index.data_ptr()))

    def step(self, n_steps=1):
        self._check(self.L.rem2d_world_step(self.h, int(n_steps), self._stream()))

    def step_ex(self, n_steps, dt, vel_iters, pos_iters):
        self._check(self.L.rem2d_world_step_ex(self.h, int(n_steps), float(dt), int(vel_iters), int(pos_iters),
                                                  self._stream()))

    # ---- zero-copy state views ----
    def view(self, name):
        v = self._views.get(name)
        if v is None:
            off, cnt, dt = C.c_size_t(), C.c_size_t(), C.c_int32()
            self._check(self.L.rem2d_world_field(self.h, _lib.FIELD_ID[name], C.byref(off), C.byref(cnt),
                                                    C.byref(dt)))
            dtype = _DTYPES[dt.value]
            nb = cnt.value * (8 if dt.value == 2 else 4)
            v = self.arena[off.value:off.value + nb].view(dtype)
            Lp = self.n_envs_padded * self.lanes
            if cnt.value == Lp:
                v = v.view(self.n_envs_padded, self.lanes)[:self.n_envs]
            elif cnt.value == Lp * self.contact_slots:
                v = v.view(self.contact_slots, self.n_envs_padded, self.lanes)[:, :self.n_envs]
            else:
                v = v[:self.n_envs]
            self._views[name] = v
        return v

    def bodies(self):
        """[n_envs, lanes, 8] = x y angle vx vy w sleepTime awake (host numpy)."""
        torch.cuda.synchronize(self.device)
        cols = [self.view(k).float() for k in ("px", "py", "ang", "vx", "vy", "w", "sleept")]
        cols.append(self.view("awake").float())
        return torch.stack(cols, dim=-1).cpu().numpy()

    def handover_failures(self, clear=False):
        """Hand-overs of this world's step trains that have failed so far (rem2d_world_handover_failures: a word in pinned host
        memory, no device call, no synchronisation -- what the launches that have FINISHED reported)."""
        n = C.c_int64()
        self._check(self.L.rem2d_world_handover_failures(self.h, C.byref(n), 1 if clear else 0))
        return n.value

    def enable_timing(self, on=True):
        """on: False / 0 = off; True = on with room for 4096 timed launches between two read-backs; an int > 1 = on with
        room for that many (the event pairs are created here, two pools of that size)."""
        self._check(self.L.rem2d_world_enable_timing(self.h, int(on) if on is not True else 1))

    def step_time_ms(self):
        """(device ms, steps) of the whole kernel sequence of the env-steps since the last call (tile pipeline)."""
        t, n = C.c_double(), C.c_int64()
        self._check(self.L.rem2d_world_step_time_ms(self.h, C.byref(t), C.byref(n)))
        return t.value, n.value

    def kernel_time_ms(self):
        t, n = C.c_double(), C.c_int64()
        self._check(self.L.rem2d_world_kernel_time_ms(self.h, C.byref(t), C.byref(n)))
        return t.value, n.value
