"""Evolutionary loop around the batched stepper.

Restates the generation loop of the reference's ``run2D.run_deap`` (``REM2D_main.py:241-348``)
without DEAP (absent from the image): tournament selection of size 4 (``tools.selTournament``),
clone, per-individual mutation, evaluation of the whole offspring batch, per-generation
min/max/mean, elite pickle.  The only change is *where* evaluation runs: ``toolbox.map(evaluate,
offspring)`` over a process pool becomes one batched episode on the GPU
(:func:`gym_rem2d_amd.evaluate.evaluate_population`), optionally sharded over the ranks of a
``torch.distributed`` job with a single fitness all-gather per generation.  Selection and mutation
run replicated on every rank from the gathered fitness with a shared RNG seed, so no genome ever
crosses a rank boundary.
"""
import copy
import os
import pickle
import random
import time

import numpy as np

from .encodings import DirectEncoding, LSystem, NNEncoding
from .modules import get_module_list

DEFAULTS = dict(population_size=100, n_evaluations=10000, mutation_prob=0.01, morphmutation_prob=0.01,
                mutation_sigma=0.1, max_depth=7, max_size=40, encoding="lsystem", checkpoint_frequency=10)


class Individual:
    """``REM2D_main.Individual`` (:84-139): a genome + fitness."""

    def __init__(self):
        self.genome = None
        self.fitness = 0
        self.tree_depth = 8

    @staticmethod
    def random(moduleList=None, config=None, encoding="lsystem"):
        self = Individual()
        if moduleList is None:
            moduleList = get_module_list()
        if config is not None:
            encoding = config["encoding"]["type"]
            self.tree_depth = int(config["morphology"]["max_depth"])
        if encoding == "direct":
            self.genome = DirectEncoding(moduleList, config)
        elif encoding == "lsystem":
            self.genome = LSystem(moduleList, config)
        elif encoding == "cppn":
            # neat-python is absent from the image: the genome is a synthetic feed-forward CPPN
            self.genome = NNEncoding(moduleList, config=config)
        else:
            raise Exception("encoding %r is not available in this build (supported: 'direct', 'lsystem', 'cppn'); "
                            "'ce' (cellular encoding) is genotype-only code outside the accelerated path" % encoding)
        self.genome.create(self.tree_depth)
        return self

    def mutate(self, MORPH_MUTATION_RATE, MUTATION_RATE, MUT_SIGMA):
        self.genome.mutate(MORPH_MUTATION_RATE, MUTATION_RATE, MUT_SIGMA)


def sel_tournament(individuals, k, tournsize=4, rng=random):
    """deap.tools.selTournament on the plain float ``fitness`` attribute the reference uses."""
    chosen = []
    for _ in range(k):
        aspirants = [rng.choice(individuals) for _ in range(tournsize)]
        chosen.append(max(aspirants, key=lambda ind: ind.fitness))
    return chosen


def make_config(**kw):
    """dict-of-dicts shaped like the reference's .cfg sections (0.cfg)."""
    p = dict(DEFAULTS)
    p.update(kw)
    return {"ea": {"batch_size": p["population_size"], "n_evaluations": p["n_evaluations"],
                   "mutation_prob": p["mutation_prob"], "morphmutation_prob": p["morphmutation_prob"],
                   "mutation_sigma": p["mutation_sigma"]},
            "morphology": {"max_depth": p["max_depth"], "max_size": p["max_size"]},
            "encoding": {"type": p["encoding"]},
            "experiment": {"checkpoint_frequency": p["checkpoint_frequency"]}}


def run_ea(config=None, population=None, evaluate_batch=None, seed=None, save_dir=None, n_generations=None,
           log=print, fitness_data=None):
    """Run the generational loop.  ``evaluate_batch(list[Individual]) -> list[float]`` defaults to one
    batched GPU episode.  Returns (population, history) with history rows (gen, min, max, mean, seconds).

    With ``save_dir`` the run leaves the reference's files in the reference's pickle format
    (REM2D_main.py:192-194,310-329; compat.dump_reference_pickle): ``s_`` (FitnessData: percentiles per generation),
    ``s_pop<gen>`` every ``checkpoint_frequency`` generations and ``s_elite<gen>`` whenever the best fitness is
    positive -- the reference can resume from them or replay the elite (REM2D_main.py:165,177-178)."""
    from .compat import FitnessData, dump_reference_pickle
    fitness_data = fitness_data if fitness_data is not None else FitnessData()
    config = config or make_config()
    if seed is not None:
        random.seed(seed)
    pop_size = int(config["ea"]["batch_size"])
    mrate = float(config["ea"]["mutation_prob"])
    morph_rate = float(config["ea"]["morphmutation_prob"])
    sigma = float(config["ea"]["mutation_sigma"])
    tree_depth = int(config["morphology"]["max_depth"])
    ckpt = int(config["experiment"]["checkpoint_frequency"])
    if n_generations is None:
        n_generations = 1 + int(int(config["ea"]["n_evaluations"]) / pop_size)
    if evaluate_batch is None:
        from .evaluate import evaluate_population
        from .env import BatchedModular2D
        from . import _lib
        env = BatchedModular2D(flags=_lib.FLAG_CONTINUOUS | _lib.FLAG_SKIP_FROZEN)

        def evaluate_batch(inds):
            # one out-of-domain creature (contacts beyond even the wide build) must not abort a generation: it gets
            # evaluate.UNRESOLVED_FITNESS, a warning names it, run_ea.last_unresolved keeps the indices per generation
            fits = evaluate_population(inds, tree_depth=tree_depth, env=env, on_error="penalty")
            run_ea.last_unresolved.append(list(getattr(env, "last_unresolved", [])))
            return fits

    run_ea.last_unresolved = []
    if population is None:
        population = [Individual.random(config=config) for _ in range(pop_size)]
        for ind, fit in zip(population, evaluate_batch(population)):
            ind.fitness = fit
    history = []
    for gen in range(n_generations):
        t0 = time.time()
        offspring = sel_tournament(population, len(population))
        offspring = [copy.deepcopy(o) for o in offspring]
        for o in offspring:
            o.mutate(morph_rate, mrate, sigma)
            o.fitness = 0
        fits = evaluate_batch(offspring)
        for ind, fit in zip(offspring, fits):
            ind.fitness = fit
        population = offspring
        row = (gen + 1, float(np.min(fits)), float(np.max(fits)), float(np.mean(fits)), time.time() - t0)
        history.append(row)
        if log:
            log("Generation %d evaluated ( %.2fs ) : Min %s, Max %s, Avg %s" % (row[0], row[4], row[1], row[2], row[3]))
        fitness_data.addFitnessData(fits, gen)
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
            prefix = os.path.join(save_dir, "s_")            # SAVE_FILE_DIRECTORY (REM2D_main.py:194)
            if gen % ckpt == 0 or gen == n_generations - 1:
                fitness_data.save(prefix)
                dump_reference_pickle(population, prefix + "pop%d" % gen)
            best = max(population, key=lambda ind: ind.fitness)
            if best.fitness > 0.0:
                dump_reference_pickle(best, prefix + "elite%d" % gen)
    run_ea.last_fitness_data = fitness_data
    return population, history
