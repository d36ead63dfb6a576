"""Network encoding: a function (CPPN) queried per connection site grows the tree.

Mirror of the tree-expansion logic of the reference's ``Encodings/Network_Encoding.py:42-222``
(``update`` / ``iterate`` / ``create`` / ``recursiveNodeGen``): inputs are (normalised depth, parent
module type, connection-site sign), outputs 0..8 decide whether a child exists, its module type, its
shape (``setMorph``) and its controller (``setControl``).

The reference evaluates a neat-python CPPN (``NeuralNetwork/NEAT_NN.py``) or its cellular-encoding
network; neat-python is pinned by the reference but absent from this image, so the *genome* here is a
synthetic fixed-topology feed-forward CPPN (:class:`FeedForwardCPPN`) -- "parity unpinned" for the
genome, exact for the expansion rules.  Input generator of BASELINE config 4.
"""
import copy
import math
import random

from ..tree import Node, Tree

MAX_MODULES = 20


class FeedForwardCPPN:
    """3 -> hidden -> 10 network with per-node activation functions drawn from the usual CPPN set."""
    ACTS = (math.tanh, math.sin, lambda x: math.exp(-x * x) * 2 - 1, lambda x: max(-1.0, min(1.0, x)))

    def __init__(self, n_inputs=3, n_outputs=10, n_hidden=8, rng=random):
        self.n_inputs, self.n_outputs, self.n_hidden = n_inputs, n_outputs, n_hidden
        g = rng.gauss
        self.w1 = [[g(0, 1.5) for _ in range(n_inputs + 1)] for _ in range(n_hidden)]
        self.a1 = [rng.randrange(len(self.ACTS)) for _ in range(n_hidden)]
        self.w2 = [[g(0, 1.0) for _ in range(n_hidden + 1)] for _ in range(n_outputs)]

    def activate(self, x):
        h = [self.ACTS[a](sum(w * v for w, v in zip(row, list(x) + [1.0]))) for row, a in zip(self.w1, self.a1)]
        return [math.tanh(sum(w * v for w, v in zip(row, h + [1.0]))) for row in self.w2]

    def mutate(self, rate=0.2, sigma=0.3, rng=random):
        for m in (self.w1, self.w2):
            for row in m:
                for i in range(len(row)):
                    if rng.uniform(0, 1) < rate:
                        row[i] += rng.gauss(0, sigma)

    def getPhenotype(self):
        return self


class _Symbol:
    def __init__(self, index, module, module_ref):
        self.index = index
        self.parent = -1
        self.moduleRef = module_ref
        self.availableConnections = list(module.available)
        self.children = []
        self.parentConnectionSite = None
        self.handled = False
        self.module = copy.deepcopy(module)
        self.controller = None


class NNEncoding:
    def __init__(self, modulelist, network=None, config=None):
        self.moduleList = copy.deepcopy(modulelist)
        if config is not None:
            self.maxTreeDepth = int(config['morphology']['max_depth'])
            self.maxModules = int(config['morphology']['max_size'])
        else:
            self.maxTreeDepth = 7
            self.maxModules = 20
        self.nn_g = network if network is not None else FeedForwardCPPN()
        self.nn_p = None
        for mod in self.moduleList:
            mod.mutate(0.5, 0.5, 0.5)

    def _query(self, index, parent, depth):
        """One network query per free connection site of `parent` (reference ``update``)."""
        born = []
        if depth > self.maxTreeDepth or index > self.maxModules:
            return index, born
        n_types = len(self.moduleList)
        for con in parent.availableConnections:
            x = [float(1) - (float(2) * (float(depth) / float(self.maxTreeDepth))),
                 float(1) - (float(2) * (float(parent.moduleRef + 1) / float(n_types))),
                 con.value[0]]
            out = list(self.nn_p.activate(x))
            if out[0] > 0.5:
                out[1] = max(-1., min(1., out[1]))
                ref = int(((out[1] * 0.5) + 0.5) * float(n_types - 1))
                ref = max(0, min(n_types - 1, ref))
                child = _Symbol(index, self.moduleList[ref], ref)
                child.module.setMorph(out[2], out[3], out[4])
                ctrl = copy.deepcopy(self.moduleList[ref].controller)
                ctrl.setControl(out[5], out[6], out[7], out[8], self.moduleList[ref].angle)
                child.controller = ctrl
                child.parent = parent.index
                child.parentConnectionSite = con
                parent.children.append(child)
                born.append(child)
                index += 1
        return index, born

    def _grow(self, sym, index, depth):
        if not sym.handled:
            sym.handled = True
            if sym.children:
                raise Exception("if symbol was not handled it shouldn't contain children")
            index, born = self._query(index, sym, depth)
            for s in born:
                s.parent = sym.index
        else:
            for c in sym.children:
                index = self._grow(c, index, depth + 1)
        return index

    def create(self, treedepth):
        self.maxTreeDepth = treedepth
        self.nn_p = self.nn_g.getPhenotype()
        base = _Symbol(0, self.moduleList[0], -1)
        base.controller = copy.deepcopy(self.moduleList[0].controller)
        index = 1
        for _ in range(treedepth):
            index = self._grow(base, index, 0)
        self.nn_p = None
        tree = Tree(self.moduleList)
        self._emit(-1, base, tree, 0)
        return tree

    def _emit(self, parent_index, sym, tree, counter):
        if counter > MAX_MODULES:
            return counter
        node = Node(sym.index, parent_index, sym.moduleRef, sym.parentConnectionSite, sym.controller)
        node.module_ = sym.module
        tree.nodes.append(node)
        for c in sym.children:
            counter += 1
            counter = self._emit(c.parent, c, tree, counter)
        return counter

    def mutate(self, MORPH_MUTATION_RATE, MUTATION_RATE, MUT_SIGMA, TREE_DEPTH=None):
        self.nn_g.mutate()
        for mod in self.moduleList:
            mod.mutate(MORPH_MUTATION_RATE, MUTATION_RATE, MUT_SIGMA)
