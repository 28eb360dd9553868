"""Non-iterative step-size (1/L) policies for PGM.

Same classes as ``sporco.pgm.stepsize`` (sporco/pgm/stepsize.py:50-145).  The
arrays they need stay on the GPU: the policy asks the solver for the few inner
products it needs (``solver.dev`` reductions) instead of touching ndarrays.
"""

from .. import _lib

__all__ = ['StepSizePolicyBase', 'StepSizePolicyCauchy', 'StepSizePolicyBB']


class StepSizePolicyBase(object):
    """Interface: ``update(solverobj, grad)`` returns the new L."""

    def update(self, solverobj, grad=None):
        raise NotImplementedError()


class StepSizePolicyCauchy(StepSizePolicyBase):
    r"""L = <g, Hess_f g> / ||g||^2 (sporco/pgm/stepsize.py:67-92)."""

    def update(self, solverobj, grad=None):
        if grad is None:
            grad = solverobj.grad_f()
        den = solverobj.dev.pair_stats(grad)[2]                      # ||g||^2
        num = solverobj.hess_quad(grad)                             # <g, Hess_f g>
        return num / den


class StepSizePolicyBB(StepSizePolicyBase):
    r"""Barzilai-Borwein: L = ||dg||^2 / <dx, dg> (sporco/pgm/stepsize.py:95-145);
    the previous iterate and gradient are kept in two device scratch arrays."""

    def __init__(self):
        self.have_prev = False

    def store_prev_state(self, solverobj, xvar, gvar):
        solverobj.dev.copy(solverobj.scratch(0), xvar)
        solverobj.dev.copy(solverobj.scratch(1), gvar)
        self.have_prev = True

    def update(self, solverobj, grad=None):
        if grad is None:
            grad = solverobj.grad_f()
        dev = solverobj.dev
        if self.have_prev:
            dg = solverobj.scratch(2)
            dev.lincomb(dg, 1.0, grad, -1.0, solverobj.scratch(1))
            st = dev.pair_stats(solverobj.var_x(), solverobj.scratch(0), dg)
        else:
            # reference initial state: xprv = gradprv = 0.0
            st = dev.pair_stats(solverobj.var_x(), -1, grad)
        den, num = st[1], st[3]
        L = num / den
        if L < 0.:
            L = solverobj.L
        return L

    def __getstate__(self):
        return {'have_prev': False, '_slots_filled': False}
