"""Readers and writers for the reference's checkpoints.

The reference pickles whole populations / elites of ``REM2D_main.Individual`` objects
(``REM2D_main.py:311-329``).  Those pickles name the reference's modules (``Encodings.LSystem``,
``gym_rem2D.morph.simple_module`` ...), which are not importable here.  The classes of this package keep the
reference's attribute names, so a checkpoint loads by mapping class paths -- nothing of the reference has to
be installed.  Module names are matched case-insensitively (the reference imports ``Encodings.lsystem`` on a
case-insensitive file system and ``Encodings.LSystem`` elsewhere).

Writing goes the other way (:func:`dump_reference_pickle`): objects of this package are pickled under the
reference's class paths (the spelling found in checkpoints written by the reference's own classes), so that
``pickle.load`` inside the reference -- ``REM2D_main.py:165,177-178`` resuming a run, ``Demo3_Evaluate_Best_Individual``
-- gets its own ``Individual`` / ``LSystem`` / ``Standard2D`` ... instances.  :class:`FitnessData` is the reference's
progress record (``DataAnalysis.py:39-56``) with the same attributes and file convention.
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
        ("dataanalysis", "FitnessData"): FitnessData,
    }


def _reference_paths():
    """class of this package -> (module, name) as the reference's own pickles spell it."""
    from . import ea, modules, tree
    from .controller import Controller
    from .encodings import direct, lsystem
    return {
        ea.Individual: ("REM2D_main", "Individual"),
        Encoding_Type: ("REM2D_main", "Encoding_Type"),
        lsystem.LSystem: ("Encodings.lsystem", "LSystem"),
        lsystem.Rule: ("Encodings.lsystem", "Rule"),
        lsystem.Symbol: ("Encodings.lsystem", "C_Module"),
        direct.DirectEncoding: ("Encodings.direct_encoding", "DirectEncoding"),
        direct.DirectTree: ("Encodings.direct_encoding", "DirectTree"),
        direct.DirectNode: ("Encodings.direct_encoding", "DirectNode"),
        tree.Tree: ("Tree", "Tree"),
        tree.Node: ("Tree", "Node"),
        modules.Standard2D: ("gym_rem2D.morph.simple_module", "Standard2D"),
        modules.BoxConnection: ("gym_rem2D.morph.simple_module", "Connection"),
        modules.Circular2D: ("gym_rem2D.morph.circular_module", "Circular2D"),
        modules.CircleConnection: ("gym_rem2D.morph.circular_module", "Connection"),
        Controller: ("Controller.m_controller", "Controller"),
        FitnessData: ("DataAnalysis", "FitnessData"),
    }


class FitnessData:
    """``DataAnalysis.FitnessData`` (DataAnalysis.py:39-56): percentiles and mean of the fitness per generation,
    pickled to ``<save dir>/s_`` by the reference's loop (REM2D_main.py:310-314)."""

    def __init__(self):
        self.p_0, self.p_25, self.p_50, self.p_75, self.p_100 = [], [], [], [], []
        self.avg = []
        self.divValues = []

    def save(self, saveFile, num=''):
        dump_reference_pickle(self, saveFile + str(num))

    def addFitnessData(self, fitnesses, gen):
        import numpy as np
        self.avg.append(np.average(fitnesses))
        self.p_0.append(np.percentile(fitnesses, 0))
        self.p_25.append(np.percentile(fitnesses, 25))
        self.p_50.append(np.percentile(fitnesses, 50))
        self.p_75.append(np.percentile(fitnesses, 75))
        self.p_100.append(np.percentile(fitnesses, 100))


class Opaque:
    """Stands in for helper objects of the reference's legacy 3D package that 2D modules still carry along
    (``Circular2D.orientation`` is a ``gym_rem.utils.Rot``, circular_module.py:37); the 2D path never reads them.
    One subclass per reference class, remembering its path so that a checkpoint can be written back unchanged."""
    _ref_path = None


_OPAQUE = {}


def _opaque_class(module, name):
    key = (module, name)
    if key not in _OPAQUE:
        _OPAQUE[key] = type(name, (Opaque,), {"_ref_path": key, "__module__": __name__})
    return _OPAQUE[key]


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
            return _opaque_class(module, name)
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


class ReferencePickler(pickle._Pickler):
    """pure-Python pickler whose class references carry the reference's module paths."""

    def __init__(self, file, protocol=2):
        super().__init__(file, protocol=protocol)
        self._paths = _reference_paths()

    def save_global(self, obj, name=None):
        path = self._paths.get(obj)
        if path is None and isinstance(obj, type) and issubclass(obj, Opaque) and obj._ref_path:
            path = obj._ref_path
        if path is None:
            mod = getattr(obj, "__module__", "") or ""
            if mod.split(".")[0] == "gym_rem2d_amd":
                raise pickle.PicklingError("%s.%s has no counterpart in the reference: it cannot go into a "
                                           "reference-format checkpoint" % (mod, getattr(obj, "__qualname__", obj)))
            return super().save_global(obj, name)
        self.write(pickle.GLOBAL + path[0].encode("utf-8") + b"\n" + path[1].encode("utf-8") + b"\n")
        self.memoize(obj)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[type] = save_global


def dump_reference_pickle(obj, path_or_file, protocol=2):
    """Pickle a population / elite / FitnessData of this package under the reference's class paths
    (REM2D_main.py:311-329).  Protocol 2 like the fixtures the reference's own classes produced."""
    if hasattr(path_or_file, "write"):
        ReferencePickler(path_or_file, protocol).dump(obj)
        return
    with open(path_or_file, "wb") as f:
        ReferencePickler(f, protocol).dump(obj)


def dumps_reference_pickle(obj, protocol=2):
    buf = io.BytesIO()
    ReferencePickler(buf, protocol).dump(obj)
    return buf.getvalue()
