"""Terrain profile generator.

Restates ``Modular2D._generate_terrain`` (``gym_rem2D/envs/Modular2DEnv.py:188-310``) and the
gym-0.18 seeding it draws from (``seed`` ``:171-173`` -> ``gym.utils.seeding.np_random``,
gym==0.18.0 pinned in ``/root/reference/requirements.txt:6``; third-party, restated from its
published algorithm).  Every evaluation re-seeds with 4 before ``reset``
(``REM2D_main.py:358``), so one terrain is shared by the whole batch: it is generated once on
the host and uploaded once (``rem2d_world_set_terrain``).

Output: 200 polyline points (199 edge bodies, friction 2.5) plus, in hardcore mode, the
static boxes of pits / stumps / stairs in creation order.
"""
import hashlib
import struct

import numpy as np

FPS = 50
SCALE = 30.0
VIEWPORT_W, VIEWPORT_H = 800, 600
TERRAIN_STEP = 14 / SCALE
TERRAIN_LENGTH = 200
TERRAIN_HEIGHT = VIEWPORT_H / SCALE / 4
TERRAIN_GRASS = 10
TERRAIN_STARTPAD = 20
FRICTION = 2.5
MAX_PERTURBANCE_TERRAIN = 24


def _bigint_from_bytes(b):
    pad = 4 - len(b) % 4
    b += b"\0" * pad
    n = len(b) // 4
    return sum(2 ** (32 * i) * v for i, v in enumerate(struct.unpack("%dI" % n, b)))


def hash_seed(seed, max_bytes=8):
    """gym.utils.seeding.hash_seed (gym 0.18)."""
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(h[:max_bytes])


def np_random(seed):
    """gym.utils.seeding.np_random (gym 0.18): RandomState seeded with the 32-bit limbs of
    sha512(str(seed))[:8]."""
    if seed is None:
        seed = int.from_bytes(np.random.bytes(8), "little")
    if not (isinstance(seed, int) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not %r" % (seed,))
    seed = seed % 2 ** 64
    big = hash_seed(seed)
    limbs = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        limbs.append(mod)
    rng = np.random.RandomState()
    rng.seed(limbs or [0])
    return rng, seed


class TerrainProfile:
    """xs, ys: float64 polyline (narrowed to binary32 at upload, like ``edgeShape(vertices=)``);
    polys: [n][4][2] float64 static boxes in creation order."""

    def __init__(self, xs, ys, polys, friction=FRICTION):
        self.xs = np.asarray(xs, dtype=np.float64)
        self.ys = np.asarray(ys, dtype=np.float64)
        self.polys = np.asarray(polys, dtype=np.float64).reshape(-1, 4, 2)
        self.friction = friction

    def f32(self):
        return (self.xs.astype(np.float32), self.ys.astype(np.float32), self.polys.astype(np.float32))


def generate_terrain(rng, hardcore=False, max_perturbance=MAX_PERTURBANCE_TERRAIN):
    """State machine of ``_generate_terrain``; consumes ``rng`` exactly like the reference
    (``max_perturbance=0`` is BASELINE.json's synthetic 'flat terrain': y == 5.0 exactly)."""
    GRASS, STUMP, PIT, STAIRS, N_STATES = range(5)
    state, velocity, y = GRASS, 0.0, TERRAIN_HEIGHT
    original_y = TERRAIN_HEIGHT
    counter, oneshot = TERRAIN_STARTPAD, False
    xs, ys, polys = [], [], []
    stair_height = stair_width = stair_steps = 0
    step = TERRAIN_STEP
    for i in range(TERRAIN_LENGTH):
        x = i * step
        xs.append(x)
        if state == GRASS and not oneshot:
            velocity = 0.5 * velocity + 0.01 * np.sign(TERRAIN_HEIGHT - y)
            if i > TERRAIN_STARTPAD:
                span = max_perturbance / TERRAIN_LENGTH * i
                velocity += rng.uniform(-span, span) / SCALE
            y += velocity
        elif state == PIT and oneshot:
            counter = rng.randint(3, 5)
            box = [(x, y), (x + step, y), (x + step, y - 4 * step), (x, y - 4 * step)]
            polys.append(box)
            polys.append([(p[0] + step * counter, p[1]) for p in box])
            counter += 2
            original_y = y
        elif state == PIT and not oneshot:
            y = original_y
            if counter > 1:
                y -= 4 * step
        elif state == STUMP and oneshot:
            counter = rng.randint(1, 3)
            polys.append([(x, y), (x + counter * step, y), (x + counter * step, y + counter * step),
                          (x, y + counter * step)])
        elif state == STAIRS and oneshot:
            stair_height = +1 if rng.rand() > 0.5 else -1
            stair_width = rng.randint(4, 5)
            stair_steps = rng.randint(3, 5)
            original_y = y
            for s in range(stair_steps):
                polys.append([(x + (s * stair_width) * step, y + (s * stair_height) * step),
                              (x + ((1 + s) * stair_width) * step, y + (s * stair_height) * step),
                              (x + ((1 + s) * stair_width) * step, y + (-1 + s * stair_height) * step),
                              (x + (s * stair_width) * step, y + (-1 + s * stair_height) * step)])
            counter = stair_steps * stair_width
        elif state == STAIRS and not oneshot:
            s = stair_steps * stair_width - counter - stair_height
            n = s / stair_width
            y = original_y + (n * stair_height) * step
        oneshot = False
        ys.append(y)
        counter -= 1
        if counter == 0:
            counter = rng.randint(int(TERRAIN_GRASS / 2), TERRAIN_GRASS)
            if state == GRASS and hardcore:
                state = rng.randint(1, N_STATES)
            else:
                state = GRASS
            oneshot = True
    return TerrainProfile(xs, ys, polys)


def make_terrain(seed=4, hardcore=False, flat=False):
    """Terrain of ``env.seed(seed); env.reset(...)``."""
    rng, _ = np_random(seed)
    return generate_terrain(rng, hardcore=hardcore, max_perturbance=0 if flat else MAX_PERTURBANCE_TERRAIN)
