"""Backtracking line searches for PGM (estimate L so that F <= Q_L).

Same classes and parameters as ``sporco.pgm.backtrack``
(BacktrackStandard sporco/pgm/backtrack.py:50-117, BacktrackRobust :120-208).
All X-sized work (proximal steps, differences, inner products) is done by the
solver's device calls; this module only compares scalars.
"""

import numpy as np

from .. import _lib

__all__ = ['BacktrackBase', 'BacktrackStandard', 'BacktrackRobust']


class BacktrackBase(object):
    """Interface: ``update(solverobj)`` performs one backtracked PGM step."""

    def update(self, solverobj):
        raise NotImplementedError()


def _quadratic_model(solverobj, gradY):
    """Q_L(x, y) = f(y) + <x - y, grad f(y)> + (L/2)||x - y||^2 in the
    unnormalised DFT domain (pgm.py:886-894)."""
    st = solverobj.dev.pair_stats(solverobj.var_x(), solverobj.var_y(), gradY)
    return solverobj.obfn_f(solverobj.var_y()) + st[1] + (solverobj.L / 2.) * st[2]


class BacktrackStandard(BacktrackBase):
    """Beck-Teboulle backtracking: multiply L by ``gamma_u`` until F <= Q."""

    def __init__(self, gamma_u=1.2, maxiter=50):
        self.gamma_u = gamma_u
        self.maxiter = maxiter

    def update(self, solverobj):
        gradY = solverobj.grad_f()
        it = 0
        while True:
            solverobj.xstep(gradY)
            f = solverobj.obfn_f(solverobj.var_x())
            Q = _quadratic_model(solverobj, gradY)
            it += 1
            if f <= Q or it >= self.maxiter:
                if f > Q:
                    solverobj.L *= self.gamma_u
                break
            solverobj.L *= self.gamma_u
        solverobj.F, solverobj.Q, solverobj.iterBTrack = f, Q, it
        solverobj.ystep()


class BacktrackRobust(BacktrackBase):
    """Florea-Vorobyov robust backtracking: L may also decrease (``gamma_d``)."""

    def __init__(self, gamma_d=0.9, gamma_u=2.0, maxiter=50):
        self.gamma_d = gamma_d
        self.gamma_u = gamma_u
        self.maxiter = maxiter
        self.Tk = 0.
        self.have_z = False

    def update(self, solverobj):
        dev = solverobj.dev
        Z = solverobj.scratch(2)
        if not self.have_z:
            dev.copy(Z, solverobj.var_x())
            self.have_z = True
        solverobj.L *= self.gamma_d
        it = 0
        while True:
            t = float(1. + np.sqrt(1. + 4. * solverobj.L * self.Tk)) / (2. * solverobj.L)
            T = self.Tk + t
            # y = (Tk * xprv + t * Z) / T
            dev.lincomb(solverobj.var_y(), self.Tk / T, solverobj.var_xprv(), t / T, Z)
            solverobj.invalidate(solverobj.var_y())
            gradY = solverobj.xstep()
            f = solverobj.obfn_f(solverobj.var_x())
            Q = _quadratic_model(solverobj, gradY)
            it += 1
            if f <= Q or it >= self.maxiter:
                if f > Q:
                    solverobj.L *= self.gamma_u
                break
            solverobj.L *= self.gamma_u
        self.Tk = T
        # Z += t L (x - y)
        tl = t * float(solverobj.L)
        dev.lincomb(Z, 1.0, Z, tl, solverobj.var_x(), -tl, solverobj.var_y())
        solverobj.F, solverobj.Q, solverobj.iterBTrack = f, Q, it

    def __getstate__(self):
        d = self.__dict__.copy()
        d['have_z'] = False
        return d
