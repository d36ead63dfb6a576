"""L-system encoding: one rewriting rule per module type.

Behavioural mirror of the reference's ``Encodings/LSystem.py:18-215``;
consumes the global ``random`` stream in the same order, so seeded individuals match the
reference tree for tree (pinned by tests/golden/layout_lsystem.json).  Used as the input
generator of BASELINE config 3 (65 536 random L-system creatures).
"""
import copy
import random

from ..tree import FastCopy, Node, Tree


class Symbol(FastCopy):
    """Placeholder for one module occurrence while the string is rewritten."""

    def __init__(self, index, module, module_ref):
        self.index = index
        self.parent = -1
        self.moduleRef = module_ref
        self.availableConnections = list(module.available)
        self.children = []
        self.theta = -1
        self.parentConnectionSite = None
        self.handled = False


class Rule(FastCopy):
    """A := A[children]; the left-hand symbol is kept and its products are attached to it."""

    def __init__(self, module_ref, module_list):
        self.moduleList = module_list
        self.moduleRef = module_ref
        self.module = Symbol(-1, module_list[module_ref], module_ref)
        self.max_children = len(self.module.availableConnections)
        self.n_children = random.randint(0, self.max_children)
        for _ in range(self.n_children):
            self._attach_random_product()

    def _attach_random_product(self):
        site = random.choice(self.module.availableConnections)
        ref = random.choice(range(len(self.moduleList)))
        product = Symbol(-1, self.moduleList[ref], ref)
        product.theta = random.randint(0, 3)
        self.module.availableConnections.remove(site)
        product.parentConnectionSite = site
        self.module.children.append(product)

    def mutate(self, MORPH_MUTATIONRATE, MUTATION_RATE, MUT_SIGMA):
        self.moduleList[self.moduleRef].mutate(MORPH_MUTATIONRATE, MUTATION_RATE, MUT_SIGMA)
        if random.uniform(0.0, 1.0) < MORPH_MUTATIONRATE:
            if self.n_children < self.max_children - 1:
                self.n_children += 1
                self._attach_random_product()
        if random.uniform(0.0, 1.0) < MORPH_MUTATIONRATE:
            if self.n_children > 0:
                self.n_children -= 1
                victim = random.choice(self.module.children)
                self.module.availableConnections.append(victim.parentConnectionSite)
                self.module.children.remove(victim)

    def update(self, index):
        out = []
        for c in self.module.children:
            index += 1
            fresh = copy.deepcopy(c)
            fresh.children = []
            fresh.index = index
            fresh.handled = False
            out.append(fresh)
        return index, out


class LSystem:
    def __init__(self, moduleList, config=None):
        self.moduleList = moduleList
        if config is not None:
            self.treeDepth = int(config['morphology']['max_depth'])
            self.maxModules = int(config['morphology']['max_size'])
        else:
            self.treeDepth = 8
            self.maxModules = 20
        self.rules = [Rule(i, moduleList) for i in range(len(moduleList))]

    def create(self, treedepth):
        base = copy.deepcopy(self.rules[0].module)
        base.children = []
        base.index = 0
        index = 0
        for _ in range(self.treeDepth):
            index = self._rewrite(base, index)
        tree = Tree(self.moduleList)
        self._emit(-1, base, tree, 0)
        return tree

    def _rewrite(self, sym, index):
        if index > self.maxModules:
            return index
        if not sym.handled:
            sym.handled = True
            if sym.children:
                raise Exception("if symbol was not handled it shouldn't contain children")
            index, products = self.rules[sym.moduleRef].update(index)
            for p in products:
                p.parent = sym.index
                sym.children.append(p)
        else:
            for c in sym.children:
                index = self._rewrite(c, index)
        return index

    def _emit(self, parent_index, sym, tree, counter):
        if counter > self.maxModules:
            return counter
        proto = self.moduleList[sym.moduleRef]
        node = Node(sym.index, parent_index, sym.moduleRef, sym.parentConnectionSite, copy.deepcopy(proto.controller))
        node.module_ = copy.deepcopy(proto)
        tree.nodes.append(node)
        for c in sym.children:
            counter += 1
            counter = self._emit(c.parent, c, tree, counter)
        return counter

    def mutate(self, MORPH_MUTATIONRATE, MUTATION_RATE, MUT_SIGMA):
        for m in self.moduleList:
            m.mutate(MORPH_MUTATIONRATE, MUTATION_RATE, MUT_SIGMA)
        for r in self.rules:
            r.mutate(MORPH_MUTATIONRATE, MUTATION_RATE, MUT_SIGMA)
