"""Readers for the reference's checkpoints.

The reference pickles whole populations / elites of ``REM2D_main.Individual`` objects
(``REM2D_main.py:311-329``).  Those pickles name the reference's modules (``Encodings.LSystem``,
``gym_rem2D.morph.simple_module`` ...), which are not importable here.  The classes of this package keep the
reference's attribute names, so a checkpoint loads by mapping class paths -- nothing of the reference has to
be installed.  Module names are matched case-insensitively (the reference imports ``Encodings.lsystem`` on a
case-insensitive file system and ``Encodings.LSystem`` elsewhere).
"""
import enum
import io
import pickle


class Encoding_Type(enum.Enum):   # REM2D_main.py:79-83
    DIRECT = 0
    LSYSTEM = 1
    NEURAL_NETWORK = 2
    CELLULAR_ENCODING = 3


def _class_map():
    from . import ea, modules, tree
    from .controller import Controller
    from .encodings import direct, lsystem
    return {
        ("rem2d_main", "Individual"): ea.Individual,
        ("__main__", "Individual"): ea.Individual,
        ("rem2d_main", "Encoding_Type"): Encoding_Type,
        ("__main__", "Encoding_Type"): Encoding_Type,
        ("encodings.lsystem", "LSystem"): lsystem.LSystem,
        ("encodings.lsystem", "Rule"): lsystem.Rule,
        ("encodings.lsystem", "C_Module"): lsystem.Symbol,
        ("encodings.direct_encoding", "DirectEncoding"): direct.DirectEncoding,
        ("encodings.direct_encoding", "DirectTree"): direct.DirectTree,
        ("encodings.direct_encoding", "DirectNode"): direct.DirectNode,
        ("tree", "Tree"): tree.Tree,
        ("tree", "Node"): tree.Node,
        ("gym_rem2d.morph.simple_module", "Standard2D"): modules.Standard2D,
        ("gym_rem2d.morph.simple_module", "Connection"): modules.BoxConnection,
        ("gym_rem2d.morph.circular_module", "Circular2D"): modules.Circular2D,
        ("gym_rem2d.morph.circular_module", "Connection"): modules.CircleConnection,
        ("controller.m_controller", "Controller"): Controller,
    }


class Opaque:
    """Stands in for helper objects of the reference's legacy 3D package that 2D modules still carry along
    (``Circular2D.orientation`` is a ``gym_rem.utils.Rot``, circular_module.py:37); the 2D path never reads them."""


class ReferenceUnpickler(pickle.Unpickler):
    def __init__(self, file):
        super().__init__(file)
        self._map = _class_map()

    def find_class(self, module, name):
        hit = self._map.get((module.lower(), name))
        if hit is not None:
            return hit
        top = module.split(".")[0].lower()
        if top == "gym_rem":
            return Opaque
        if top in ("encodings", "gym_rem2d", "controller", "tree", "rem2d_main", "neat"):
            raise pickle.UnpicklingError("reference class %s.%s has no counterpart in gym_rem2d_amd "
                                         "(supported: direct and L-system genomes)" % (module, name))
        return super().find_class(module, name)


def load_reference_pickle(path_or_bytes):
    """A population (list of Individual), an elite (Individual) or any other pickled reference object."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return ReferenceUnpickler(io.BytesIO(path_or_bytes)).load()
    with open(path_or_bytes, "rb") as f:
        return ReferenceUnpickler(f).load()
