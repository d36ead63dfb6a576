#!/usr/bin/env python3
"""Build container only (imports /root/reference in place, like capture_golden.py): checkpoints WRITTEN by this
package load inside the reference's own classes and express to the same trees.

  1. a population evolved here (direct + L-system genomes, mutated) is pickled with compat.dump_reference_pickle;
  2. plain pickle.load with the reference's modules on sys.path turns it into the reference's Individual / LSystem /
     DirectEncoding / Standard2D / Circular2D / Controller objects (REM2D_main.py:165,177-178 do exactly this);
  3. genome.create() of those objects -- the reference's code -- gives the same node lists, module and controller
     parameters as this package's genomes, and the reference's env builds the same bodies / joints from them;
  4. FitnessData round-trips through DataAnalysis.FitnessData.
Prints one line per check; exit code 0 iff all agree."""
import importlib.util
import io
import os
import pickle
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import capture_golden as CG  # noqa: E402

REF = CG.REF


def main():
    from gym_rem2d_amd import compat, ea
    random.seed(11)
    pop = [ea.Individual.random(encoding="lsystem" if k % 2 else "direct") for k in range(24)]
    for k, ind in enumerate(pop):
        for _ in range(k % 4):
            ind.mutate(0.4, 0.4, 0.3)
        ind.fitness = 0.25 * k
    ours = [CG.dump_tree(ind.genome.create(8)) for ind in pop]
    blob = compat.dumps_reference_pickle(pop)
    fd = compat.FitnessData()
    fd.addFitnessData([1.0, 2.0, 4.0], 0)
    fd_blob = compat.dumps_reference_pickle(fd)

    # ---- the reference side ----
    sys.path.insert(0, REF)
    CG.install_gym_stub()
    CG.install_box2d_stub()
    import matplotlib
    matplotlib.use("Agg")
    CG.alias_case_insensitive()
    import enum
    main_mod = types.ModuleType("REM2D_main")   # REM2D_main itself needs deap: its two small classes are stood in for

    class Encoding_Type(enum.Enum):
        DIRECT = 0
        LSYSTEM = 1
        NEURAL_NETWORK = 2
        CELLULAR_ENCODING = 3

    class Individual:
        def __init__(self):
            self.genome = None
            self.fitness = 0
    for c in (Encoding_Type, Individual):
        c.__module__, c.__qualname__ = "REM2D_main", c.__name__
        setattr(main_mod, c.__name__, c)
    sys.modules["REM2D_main"] = main_mod
    spec = importlib.util.spec_from_file_location("DataAnalysis", os.path.join(REF, "DataAnalysis.py"))
    DA = importlib.util.module_from_spec(spec)
    sys.modules["DataAnalysis"] = DA
    spec.loader.exec_module(DA)

    ref_pop = pickle.load(io.BytesIO(blob))
    ok = True
    mods = {type(i.genome).__module__ for i in ref_pop}
    print("unpickled %d individuals into reference classes from %s" % (len(ref_pop), sorted(mods)))
    ok &= all(m.startswith("Encodings.") for m in mods) and type(ref_pop[0]).__module__ == "REM2D_main"
    theirs = [CG.dump_tree(ind.genome.create(8)) for ind in ref_pop]
    same = theirs == ours
    print("genome.create() in the reference == in this package for all %d trees: %s" % (len(ours), same))
    ok &= same
    ok &= [i.fitness for i in ref_pop] == [i.fitness for i in pop]
    # the reference's env builds the same robot from the unpickled genome as this package does from its own
    from gym_rem2D.envs import Modular2DEnv as M
    M.COLOR_CONTROL = False
    from gym_rem2d_amd.compiler import build_creature
    import copy
    n_ok = 0
    for ind_r, ind_o in zip(ref_pop, pop):
        env = M.Modular2D()
        env.seed(4)
        env.reset(tree=ind_r.genome.create(8), module_list=ind_r.genome.moduleList)
        lay = CG.dump_layout(env, n_ctrl_steps=0)
        spec_o = build_creature(copy.deepcopy(ind_o.genome.create(8)).getNodes(), ind_o.genome.moduleList)[0]
        bodies_o = [(CG.f32(b._x), CG.f32(b._y), CG.f32(b._angle)) for b in spec_o.bodies]
        bodies_r = [(CG.f32(b["x"]), CG.f32(b["y"]), CG.f32(b["angle"])) for b in lay["bodies"]]
        n_ok += bodies_o == bodies_r and len(lay["joints"]) == len(spec_o.joints)
    print("reference env.reset on the unpickled genomes builds the same bodies as this package: %d / %d" % (n_ok, len(pop)))
    ok &= n_ok == len(pop)
    rfd = pickle.load(io.BytesIO(fd_blob))
    same_fd = type(rfd) is DA.FitnessData and rfd.avg == fd.avg and rfd.p_50 == fd.p_50
    print("FitnessData -> DataAnalysis.FitnessData:", same_fd)
    ok &= same_fd
    print("ROUNDTRIP", "OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
