"""Phenotype blueprint containers.

Mirrors the interface of the reference's ``Tree.py:5-26`` (``Tree.nodes`` /
``Tree.getNodes()``; ``Node`` fields ``index, parent, type,
parent_connection_coordinates, controller, expressed, component, module_``) so that
encodings written against the reference keep working; SURVEY.md section 1 lists it as
the morphology interface that must stay intact.
"""


class Tree:
    def __init__(self, moduleList, controller=None):
        self.nodes = []
        self.moduleList = moduleList

    def getNodes(self):
        return self.nodes


class Node:
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
