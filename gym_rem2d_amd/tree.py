"""Phenotype blueprint containers.

Mirrors the interface of the reference's ``Tree.py:5-26`` (``Tree.nodes`` /
``Tree.getNodes()``; ``Node`` fields ``index, parent, type,
parent_connection_coordinates, controller, expressed, component, module_``) so that
encodings written against the reference keep working; SURVEY.md section 1 lists it as
the morphology interface that must stay intact.
"""
import copy as _copy
import enum as _enum

_ATOMS = (int, float, str, bool, type(None), type, _enum.Enum)


class FastCopy:
    """``copy.deepcopy`` support without the generic reduce/reconstruct machinery: genomes are cloned once
    per offspring and modules/controllers once per expressed node (``LSystem.create``,
    ``REM2D_main.py:285``), which made deepcopy 70 % of the host-side genotype->phenotype time.
    Same result as the default deep copy (memo-aware, attribute by attribute)."""

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        d = new.__dict__
        for k, v in self.__dict__.items():
            if isinstance(v, _ATOMS):
                d[k] = v
            elif type(v) is tuple and all(isinstance(e, _ATOMS) for e in v):
                d[k] = v
            elif type(v) is list and all(isinstance(e, _ATOMS) for e in v):
                d[k] = list(v)
            else:
                d[k] = _copy.deepcopy(v, memo)
        return new


class Tree:
    def __init__(self, moduleList, controller=None):
        self.nodes = []
        self.moduleList = moduleList

    def getNodes(self):
        return self.nodes


class Node(FastCopy):
    def __init__(self, index, parent, type, parent_connection_coordinates, controller=None,
                 component=None, module_=None):
        self.index = index
        self.type = type
        self.parent = parent
        self.parent_connection_coordinates = parent_connection_coordinates
        self.controller = controller  # decentralised open-loop controller of this node's joint
        self.expressed = False        # set by the env when the node has been visited by create_robot
        self.component = component    # [BodyView] once a rigid body exists for the node, else None
        self.module_ = module_

    def __bool__(self):
        return self.expressed

    def __repr__(self):
        return "Node(index=%r, parent=%r, type=%r, con=%r)" % (
            self.index, self.parent, self.type, self.parent_connection_coordinates)
