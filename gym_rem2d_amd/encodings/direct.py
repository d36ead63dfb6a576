"""Direct encoding: the genome is the phenotype tree itself.

Behavioural mirror of the reference's ``Encodings/Direct_Encoding.py:7-139`` (SURVEY.md 8f
rank 1), including its iteration-while-mutating semantics, so that seeded individuals match
the reference (pinned by tests/golden/layout_direct.json).  Input generator of BASELINE
config 1 (single random direct-encoding individual).
"""
import copy
import random

from ..controller import Controller
from ..tree import Node, Tree


class DirectNode(Node):
    def __init__(self, index, parent, type, orientation, control, module_):
        super().__init__(index, parent, type, orientation, control, module_=copy.deepcopy(module_))
        self.availableConnections = self.module_.available
        self.children = []

    def addChild(self, module, index, parent, moduleRef, moduleController, parentConnectionSite):
        index += 1
        self.children.append(DirectNode(index, parent, moduleRef, parentConnectionSite, moduleController, module))
        self.availableConnections.remove(parentConnectionSite)
        return index


class DirectTree(Tree):
    def __init__(self, module_list):
        super().__init__(module_list)
        control = Controller()
        self.index = 0
        self.tree_nodes = [DirectNode(self.index, -1, 0, None, control, copy.deepcopy(module_list[0]))]

    def getNodes(self):
        out = []

        def walk(n):
            out.append(n)
            for c in n.children:
                walk(c)
        walk(self.tree_nodes[0])
        self.nodes = out
        return self.nodes


class DirectEncoding:
    def __init__(self, moduleList, config=None):
        self.moduleList = moduleList
        self.tree = DirectTree(moduleList)
        self.n_modules = 1
        if config is not None:
            self.maxDepth = int(config['morphology']['max_depth'])
            self.maxModules = int(config['morphology']['max_size'])
        else:
            self.maxDepth = 8
            self.maxModules = 20
        for _ in range(5):
            self.mutate(0.5, 0.5, 0.5)

    def create(self, treedepth):
        for node in self.tree.nodes:
            node.controller.i_state = 0
        return self.tree

    def countModules(self):
        def count(n):
            return 1 + sum(count(c) for c in n.children)
        self.n_modules = count(self.tree.tree_nodes[0])

    def mutateNode(self, node, morphMutationRate, mutationRate, sigma, depth):
        self.countModules()
        # NB both loops mutate the list they iterate (removal / addChild), as in the reference:
        # python's list iterator then skips the element that slides into the freed slot.
        for mod in node.children:
            if random.uniform(0, 1) < float(morphMutationRate) / float(2) / float(self.n_modules):
                if depth != 0:
                    node.availableConnections.append(mod.parent_connection_coordinates)
                    node.children.remove(mod)
                    self.countModules()
            else:
                self.mutateNode(mod, morphMutationRate, mutationRate, sigma, depth + 1)
        for con in node.availableConnections:
            self.countModules()
            if (self.n_modules < self.maxModules and depth < self.maxDepth
                    and random.uniform(0, 1) < morphMutationRate / float(self.n_modules)):
                ref = random.randint(0, len(self.moduleList) - 1)
                ctrl = Controller()
                self.tree.index = node.addChild(copy.deepcopy(self.moduleList[ref]), self.tree.index, node.index,
                                                ref, ctrl, con)
        node.module_.mutate(morphMutationRate, mutationRate, sigma)
        node.controller.mutate(mutationRate, sigma, node.module_.angle)

    def reassignIndices(self):
        def walk(n, i):
            n.index = i
            i += 1
            for c in n.children:
                i = walk(c, i)
                c.parent = n.index
            return i
        self.index = walk(self.tree.tree_nodes[0], 0)

    def mutate(self, morphMutationRate, mutationRate, sigma):
        self.mutateNode(self.tree.tree_nodes[0], morphMutationRate, mutationRate, sigma, 0)
        self.countModules()
        self.reassignIndices()
