"""Momentum coefficient rules for accelerated PGM (FISTA).

Same classes and constructor arguments as ``sporco.pgm.momentum``
(sporco/pgm/momentum.py:48-142); they are pure host scalars.
"""

import numpy as np

__all__ = ['MomentumBase', 'MomentumNesterov', 'MomentumLinear', 'MomentumGenLinear']


class MomentumBase(object):
    """Interface: ``update(arg)`` returns the next momentum coefficient t."""

    def update(self, *args):
        raise NotImplementedError()


class MomentumNesterov(MomentumBase):
    r"""t_{k+1} = (1 + sqrt(1 + 4 t_k^2)) / 2; ``update`` takes the current t."""

    def update(self, t):
        return 0.5 * float(1. + np.sqrt(1. + 4. * t ** 2))


class MomentumLinear(MomentumBase):
    r"""t_{k+1} = (k + b) / b; ``update`` takes the iteration number."""

    def __init__(self, b=2.):
        self.b = b

    def update(self, k):
        return (k + self.b) / self.b


class MomentumGenLinear(MomentumBase):
    r"""t_{k+1} = (k + a) / b; ``update`` takes the iteration number."""

    def __init__(self, a=50., b=2.):
        self.a = a
        self.b = b

    def update(self, k):
        return (k + self.a) / self.b
