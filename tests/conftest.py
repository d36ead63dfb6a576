import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def flat_terrain():
    from gym_rem2d_amd import make_terrain
    return make_terrain(4, flat=True)


@pytest.fixture(scope="session")
def rough_terrain():
    from gym_rem2d_amd import make_terrain
    return make_terrain(4)


def oracle_terrain(O, terrain):
    xs, ys, polys = terrain.f32()
    return O.Terrain(xs, ys, polys if len(polys) else None, terrain.friction)
