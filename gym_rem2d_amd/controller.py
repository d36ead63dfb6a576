"""Per-node open-loop sine oscillator (reference: ``Controller/m_controller.py:5-58``).

``update`` is on the hot path (evaluated for every expressed node every step,
``Modular2DEnv.py:620-623``); inside the batched stepper it runs on the GPU from the
four parameters and ``i_state`` exported by :meth:`params`.  The Python method is kept
for API parity and for host-side tests.
"""
import math
import random

from .tree import FastCopy


class Controller(FastCopy):
    MAX_AMP = 1
    MAX_PHASE = 1
    MAX_OFFSET = math.pi
    MAX_FREQ = 0.1

    def __init__(self, rng=random):
        self.i_state = 0
        self.output = 0
        # draw order matters for seeded reproducibility: amplitude, phase, frequency, offset
        self.amplitude = rng.uniform(0, self.MAX_AMP)
        self.phase = rng.uniform(-self.MAX_PHASE, self.MAX_PHASE)
        self.frequency = rng.uniform(-self.MAX_FREQ, self.MAX_FREQ)
        self.offset = rng.uniform(-self.MAX_OFFSET, self.MAX_OFFSET)

    def update(self, input):
        self.phase += input
        self.i_state += self.frequency
        self.output = (self.amplitude * (math.sin(self.i_state + self.phase))) + self.offset
        return self.output

    def params(self):
        """(amplitude, phase, frequency, offset, i_state) as python floats."""
        return (float(self.amplitude), float(self.phase), float(self.frequency), float(self.offset),
                float(self.i_state))

    def minMax(self, angle):
        self.amplitude = min(max(self.amplitude, 0), self.MAX_AMP)
        self.phase = min(max(self.phase, -self.MAX_PHASE), self.MAX_PHASE)
        self.frequency = min(max(self.frequency, -self.MAX_FREQ), self.MAX_FREQ)
        if self.offset > angle / 2:
            self.offset = angle / 2
        elif self.offset < -angle / 2:
            self.offset = -angle / 2

    def setControl(self, a, b, c, d, angle):
        self.amplitude = ((a + 1.0) * 0.5) * self.MAX_AMP
        self.phase = b * self.MAX_PHASE
        self.offset = c * self.MAX_OFFSET
        self.frequency = d * self.MAX_FREQ
        self.minMax(angle)

    def mutate(self, mutationrate, sigma, angle, rng=random):
        # NB the reference adds gauss(mean=value) to the value (m_controller.py:51-58); kept.
        if rng.uniform(0.0, 1.0) < mutationrate:
            self.amplitude += rng.gauss(self.amplitude, sigma)
        if rng.uniform(0.0, 1.0) < mutationrate:
            self.phase += rng.gauss(self.phase, sigma)
        if rng.uniform(0.0, 1.0) < mutationrate:
            self.frequency += rng.gauss(self.frequency, sigma * 0.1)
        if rng.uniform(0.0, 1.0) < mutationrate:
            self.offset += rng.gauss(self.offset, sigma)
        self.minMax(angle)
