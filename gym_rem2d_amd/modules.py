"""Morphological module types: ``Standard2D`` (box) and ``Circular2D`` (circle).

Host-side mirror of the reference's module interface
(``gym_rem2D/morph/simple_module.py:21-92,147-199,231-313``,
``circular_module.py:23-85,138-221``, ``abstract_module.py:38-53``): same attribute
names (``type``, ``width``/``height``/``radius``, ``angle``, ``torque``, ``controller``,
``connection_type``, ``available``), same ``create(...)`` and
``get_global_position_of_connection_site(...)`` call shapes.  Where the reference calls
pybox2d (``world.CreateDynamicBody`` / ``mu.create_joint``) these classes talk to a
:class:`gym_rem2d_amd.compiler.CreatureBuilder`, which records one creature's bodies and
joints for upload into the batched MI355X stepper.

All geometry is evaluated in Python doubles with the reference's operation order, and
values that pybox2d would hand back as C floats (``body.position``, ``body.angle``) are
rounded to binary32 by the builder, so the resulting anchors match the reference bit for
bit (pinned by tests/golden/layout_*.json).
"""
import math
import random
from enum import Enum

from .controller import Controller
from .tree import FastCopy


class BoxConnection(Enum):
    """Connection sites of a box module (simple_module.py:21-25)."""
    left = (-1., 0., 0.)
    right = (1., 0., 0.)
    top = (0., 1.0, 0.)


class CircleConnection(Enum):
    """Connection sites of a circle module (circular_module.py:23-27): mirrored signs."""
    left = (1., 0., 0.)
    right = (-1., 0., 0.)
    top = (0., 1.0, 0.)


class _Vec:
    """Minimal x/y(/z) record used for connection sites (Modular2DEnv.py:85-99)."""
    __slots__ = ("x", "y", "z")

    def __init__(self, x=None, y=None, z=None):
        self.x, self.y, self.z = x, y, z


class ConnectionSite:
    def __init__(self, position, orientation):
        self.position = position
        self.orientation = orientation


class Module(FastCopy):
    """Common tree helpers (abstract_module.py:23-53)."""
    connection_type = None
    _children = None

    @property
    def children(self):
        if not self.connection_type:
            return []
        return [self._children[c] for c in self.connection_type if c in self._children]

    @property
    def available(self):
        if not self.connection_type:
            return []
        return [c for c in self.connection_type if c not in self._children]


def _site_to_anchor_frame(site, parent_body, child_body):
    """Local anchors of the revolute joint (module_utility.py:7-17).

    ``parent_body``/``child_body`` expose ``position`` and ``angle`` already rounded to
    binary32, exactly what pybox2d returns to the reference at this point."""
    sx, sy, sth = site.position.x, site.position.y, site.orientation.x
    dis_a = math.sqrt(math.pow(sx - parent_body.position.x, 2) + math.pow(sy - parent_body.position.y, 2))
    ang_a = sth - parent_body.angle + math.pi / 2
    anchor_a = (math.cos(ang_a) * dis_a, math.sin(ang_a) * dis_a)
    dis_b = math.sqrt(math.pow(sx - child_body.position.x, 2) + math.pow(sy - child_body.position.y, 2))
    ang_b = child_body.angle - sth - math.pi / 2
    anchor_b = (math.cos(ang_b) * dis_b, math.sin(ang_b) * dis_b)
    return anchor_a, anchor_b


class Standard2D(Module):
    """Box module with one actuated revolute joint to its parent."""

    MAX_HEIGHT, MIN_HEIGHT = 1.0, 0.5
    MAX_WIDTH, MIN_WIDTH = 1.0, 0.5
    MAX_ANGLE, MIN_ANGLE = math.pi, 0

    def __init__(self, theta=0, size=(0.1, 0.1, 0.0), rng=random):
        assert len(size) == 3, "Size must be a 3 element vector"
        self.theta = theta % 2
        self.size = tuple(size)
        self.connection_type = BoxConnection
        self._children = {}
        self.controller = Controller(rng)
        # un-clamped defaults until the first mutate()/setMorph() (simple_module.py:41-43)
        self.width = 0.2
        self.height = 0.8
        self.angle = math.pi / 2
        self.type = "SIMPLE"
        self.torque = 50

    def limitWH(self):
        self.height = min(max(self.height, self.MIN_HEIGHT), self.MAX_HEIGHT)
        self.width = min(max(self.width, self.MIN_WIDTH), self.MAX_WIDTH)
        self.angle = min(max(self.angle, self.MIN_ANGLE), self.MAX_ANGLE)

    def mutate(self, MORPH_MUTATION_RATE, MUTATION_RATE, MUT_SIGMA, rng=random):
        if rng.uniform(0, 1) < MORPH_MUTATION_RATE:
            self.width = rng.gauss(self.width, MUT_SIGMA)
        if rng.uniform(0, 1) < MORPH_MUTATION_RATE:
            self.height = rng.gauss(self.height, MUT_SIGMA)
        if rng.uniform(0, 1) < MORPH_MUTATION_RATE:
            self.angle = rng.gauss(self.angle, MUT_SIGMA * math.pi)
        self.limitWH()
        if self.controller:
            self.controller.mutate(MUTATION_RATE, MUT_SIGMA, self.angle, rng)

    def setMorph(self, val1, val2, val3):
        # the reference derives both width and height from val1 (simple_module.py:89-90)
        self.width = (val1 * 0.5 * (self.MAX_WIDTH - self.MIN_WIDTH)) + 0.5 * (self.MAX_WIDTH - self.MIN_WIDTH)
        self.height = (val1 * 0.5 * (self.MAX_HEIGHT - self.MIN_HEIGHT)) + 0.5 * (self.MAX_HEIGHT - self.MIN_HEIGHT)
        self.angle = self.MIN_ANGLE + (((val3 + 1.0) * 0.5) * (self.MAX_ANGLE - self.MIN_ANGLE))
        self.limitWH()

    def get_angle(self, add_angle=0.0, con=None):
        return self.angle * con.value[0] if con is not None else add_angle

    def get_global_position_of_connection_site(self, con=None, parent_component=None):
        """Where a ray from the box centre at local angle ``con*angle + pi/2`` leaves the box,
        in world coordinates, and the orientation a child attached there gets
        (simple_module.py:147-199)."""
        if con is None:
            con = BoxConnection.left
        if parent_component is None:
            raise Exception("Parent component is none...")
        ray = con.value[0] * self.angle + math.pi / 2
        while ray > 2 * math.pi:
            ray -= 2 * math.pi
        flip_x = -1. if 0.5 * math.pi < ray < 1.5 * math.pi else 1.
        flip_y = -1. if math.pi < ray < 2 * math.pi else 1.
        s, c = math.sin(ray), math.cos(ray)
        if 2 * s == 0:
            hit_tb = (10000, 10000)          # ray parallel to the top/bottom faces
        else:
            hit_tb = ((self.height * c) / (2 * s) * flip_y, self.height / 2 * flip_y)
        if 2 * c == 0:
            hit_lr = (10000, 10000)          # ray parallel to the side faces
        else:
            hit_lr = (self.width / 2 * flip_x, (self.width * s) / (2 * c) * flip_x)
        d_tb = math.sqrt(math.pow(hit_tb[0], 2) + math.pow(hit_tb[1], 2))
        d_lr = math.sqrt(math.pow(hit_lr[0], 2) + math.pow(hit_lr[1], 2))
        reach = d_lr if d_lr < d_tb else d_tb
        pangle = parent_component.angle
        ppos = parent_component.position
        world_pos = [(math.cos(pangle + ray) * reach) + ppos[0], (math.sin(pangle + ray) * reach) + ppos[1]]
        return world_pos, pangle + ray - math.pi / 2

    def create(self, world, TERRAIN_HEIGHT, module=None, node=None, connection_site=None, p_c=None,
               module_list=None, position=None):
        """Emit this box (and its joint to ``p_c``) into the builder ``world``
        (simple_module.py:231-313).  Returns ``(components, joints)``; both empty when the
        module would start inside the terrain (``:268-271``)."""
        if p_c is not None and connection_site is None:
            raise Exception("a connection_site is needed to attach a component to a parent component")
        n_height, n_width, angle = 0.5, 0.5, 0
        if node is not None:
            src = node.module_ if node.module_ is not None else module_list[node.type]
            n_height, n_width = src.height, src.width
        elif module is not None:
            n_height, n_width = module.height, module.width
            angle = module.get_angle(0)
        pos = position if position is not None else [7, 7, 0]
        if p_c is not None:
            up = connection_site.orientation.x + angle + math.pi / 2
            pos[0] = (math.cos(up) * n_height / 2) + connection_site.position.x
            pos[1] = (math.sin(up) * n_height / 2) + connection_site.position.y
        if pos[1] - math.sqrt(math.pow(n_width, 2) + math.pow(n_height, 2)) < TERRAIN_HEIGHT:
            if node is not None:
                node.component = None
            return [], []
        if connection_site:
            angle += connection_site.orientation.x
        body = world.add_box(n_width / 2, n_height / 2, pos[0], pos[1], angle)
        if node is not None:
            node.component = [body]
        joints = []
        if connection_site is not None:
            a, b = _site_to_anchor_frame(connection_site, p_c, body)
            joints.append(world.add_revolute(p_c, body, a, b, self.torque, node))
        return [body], joints


class Circular2D(Module):
    """Circle module; always a leaf: ``connection_type`` is never set
    (circular_module.py:31-53), so ``available`` is empty."""

    MIN_RADIUS, MAX_RADIUS = 0.25, 0.5
    MIN_ANGLE, MAX_ANGLE = math.pi / 4, math.pi * 2

    def __init__(self, theta=0, size=(0.1, 0.1, 0.0), rng=random):
        assert len(size) == 3, "Size must be a 3 element vector"
        self.theta = theta % 2
        self.size = tuple(size)
        self._children = {}
        self.controller = Controller(rng)
        self.radius = 0.25
        self.angle = math.pi / 2
        self.type = "CIRCLE"
        self.torque = 50

    def limitWH(self):
        self.radius = min(max(self.radius, self.MIN_RADIUS), self.MAX_RADIUS)
        self.angle = min(max(self.angle, self.MIN_ANGLE), self.MAX_ANGLE)

    def mutate(self, MORPH_MUTATION_RATE, MUTATION_RATE, MUT_SIGMA, rng=random):
        if rng.uniform(0, 1) < MORPH_MUTATION_RATE:
            self.radius = rng.gauss(self.radius, MUT_SIGMA)
        if rng.uniform(0, 1) < MORPH_MUTATION_RATE:
            self.angle = rng.gauss(self.angle, MUT_SIGMA * math.pi)
        self.limitWH()
        if self.controller is not None:
            self.controller.mutate(MUTATION_RATE, MUT_SIGMA, self.angle, rng)

    def setMorph(self, val1, val2, val3):
        self.radius = val1 + 1.5
        self.angle = self.MIN_ANGLE + (((val3 + 1.0) * 0.5) * (self.MAX_ANGLE - self.MIN_ANGLE))
        self.limitWH()

    def get_global_position_of_connection_site(self, con=None, parent_component=None):
        """Point on the rim at ``con*angle`` from 'up' (circular_module.py:138-155)."""
        if con is None:
            con = CircleConnection.left
        turn = con.value[0] * self.angle
        if parent_component:
            turn += parent_component.angle
        rim = [math.cos(turn + math.pi / 2) * self.radius, math.sin(turn + math.pi / 2) * self.radius]
        if parent_component is None:
            return rim, turn
        return [rim[0] + parent_component.position[0], rim[1] + parent_component.position[1]], turn

    def create(self, world, TERRAIN_HEIGHT, module=None, node=None, connection_site=None, p_c=None,
               module_list=None, position=None):
        """Emit this circle (circular_module.py:157-221); skipped when ``y - r`` is below the
        terrain height (``:186-189``)."""
        if p_c is not None and connection_site is None:
            raise Exception("a connection_site is needed to attach a component to a parent component")
        r = self.radius
        angle = 0
        pos = position if position is not None else [7, 10, 0]
        if p_c is not None:
            up = connection_site.orientation.x + math.pi / 2
            pos[0] = (math.cos(up) * r) + connection_site.position.x
            pos[1] = (math.sin(up) * r) + connection_site.position.y
        if connection_site:
            angle += connection_site.orientation.x
        if pos[1] - r < TERRAIN_HEIGHT:
            if node is not None:
                node.component = None
            return [], []
        body = world.add_circle(r, pos[0], pos[1], angle)
        if node is not None:
            node.component = [body]
        joints = []
        if connection_site is not None:
            a, b = _site_to_anchor_frame(connection_site, p_c, body)
            joints.append(world.add_revolute(p_c, body, a, b, self.torque, node))
        return [body], joints


def get_module_list(rng=random):
    """4 boxes + 4 circles (REM2D_main.py:69-77)."""
    return [Standard2D(rng=rng) for _ in range(4)] + [Circular2D(rng=rng) for _ in range(4)]
