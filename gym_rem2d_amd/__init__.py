"""gym_rem2d_amd -- MI355X-native batched 2D rigid-body stepper behind gym_rem2D's env API.

Scope: the ``world.Step()`` hot loop of ``Modular2D.step()/reset()``
and the ``evaluate()`` loop that drives it, as hand-written HIP kernels for gfx950 behind a
C ABI (include/rem2d.h), with the reference's env / Tree / module interface kept intact on
the host side.  Importing the package does not load the HIP library; the first use of the
stepper does, and fails loudly if it is missing.
"""
from .tree import Tree, Node  # noqa: F401
from .controller import Controller  # noqa: F401
from .modules import Standard2D, Circular2D, BoxConnection, CircleConnection, get_module_list  # noqa: F401
from .compiler import build_creature, Morphology, CreatureSpec  # noqa: F401
from .terrain import make_terrain, TerrainProfile  # noqa: F401
from .gymshim import make, register  # noqa: F401  (gym.make('Modular2DLocomotion-v0'), gym_rem2D/__init__.py:5-7)

__version__ = "0.1.0"
