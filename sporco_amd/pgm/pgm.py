"""Proximal gradient (FISTA) solver framework, host side.

Keeps the loop, Options tree, IterationStats and timers of the reference's
``sporco.pgm.pgm`` (PGM: sporco/pgm/pgm.py:34-705, PGMDFT: :708-894).  In
this backend the iterates are frequency-domain arrays resident on the GPU, so
``var_x() / var_y() / var_xprv()`` and ``grad_f()`` return *handles* (state
variable ids of the device solver) rather than ndarrays, and the momentum /
backtracking / step-size policies (sibling modules) combine them through
device reductions.
"""

import copy

import numpy as np

from .. import _lib
from .. import cdict
from .. import common
from .. import util
from .backtrack import BacktrackRobust
from .momentum import MomentumNesterov
from .stepsize import StepSizePolicyBB

__all__ = ['PGM', 'PGMDFT']


class PGM(common.IterativeSolver):
    r"""Base class: minimise f(x) + g(x), f smooth, via accelerated proximal
    gradient steps x = prox_{g/L}(y - grad f(y)/L)."""

    class Options(cdict.ConstrainedDict):
        """PGM options; keys and defaults as sporco/pgm/pgm.py:157-166."""

        defaults = {'FastSolve': False, 'Verbose': False, 'StatusHeader': True,
                    'DataType': None, 'X0': None, 'Callback': None,
                    'MaxMainIter': 1000, 'IterTimer': 'solve', 'RelStopTol': 1e-3,
                    'L': None, 'AutoStop': {'Enabled': False, 'Tau0': 1e-2},
                    'Monotone': False, 'Momentum': MomentumNesterov(),
                    'StepSizePolicy': None, 'Backtrack': None}

        def __init__(self, opt=None):
            cdict.ConstrainedDict.__init__(self, {} if opt is None else opt)

    fwiter = 4
    fpothr = 2
    itstat_fields_objfn = ('ObjFun', 'FVal', 'GVal')
    itstat_fields_alg = ('Rsdl', 'F_Btrack', 'Q_Btrack', 'IterBTrack', 'L')
    itstat_fields_extra = ()
    hdrtxt_objfn = ('Fnc', 'f', 'g')
    hdrval_objfun = {'Fnc': 'ObjFun', 'f': 'FVal', 'g': 'GVal'}

    def __new__(cls, *args, **kwargs):
        obj = super(PGM, cls).__new__(cls)
        obj.timer = util.Timer(['init', 'solve', 'solve_wo_func', 'solve_wo_rsdl',
                                'solve_wo_btrack'])
        obj.timer.start('init')
        return obj

    def __init__(self, xshape, dtype, opt=None):
        if opt is None:
            opt = PGM.Options()
        if not isinstance(opt, PGM.Options):
            raise TypeError("Parameter opt must be an instance of PGM.Options")
        self.opt = opt
        self.set_dtype(opt, dtype)
        self.set_attr('L', opt['L'], dval=1.0, dtype=self.dtype)
        # a step-size policy is ignored when backtracking is enabled (pgm.py:240-244)
        self.stepsizepolicy = None if opt['Backtrack'] is not None else opt['StepSizePolicy']
        self.momentum = opt['Momentum']
        if opt['AutoStop', 'Enabled']:
            self.tau0 = opt['AutoStop', 'Tau0']
        self.init_state(xshape)
        self.F = None
        self.Q = None
        self.iterBTrack = None
        self.backtrack = opt['Backtrack']
        self.itstat = []
        self.k = 0
        self.t = 1

    def init_state(self, xshape):
        if self.opt['X0'] is None:
            self.X = np.zeros(xshape, dtype=self.dtype)
        else:
            self.X = self.opt['X0'].astype(self.dtype, copy=True)
        self.Y = None

    def solve(self):
        """Run (or continue) the iterations; see sporco/pgm/pgm.py:284-382."""
        fmtstr, nsep = self.display_start()
        labels = ['solve', 'solve_wo_func', 'solve_wo_rsdl', 'solve_wo_btrack']
        self.timer.start(labels)
        for self.k in range(self.k, self.k + self.opt['MaxMainIter']):
            if self.fused_iteration():
                pass
            elif self.opt['Backtrack'] is not None and self.k >= 0:
                self.on_iteration_start()
                self.timer.stop('solve_wo_btrack')
                self.backtrack.update(self)
                self.timer.start('solve_wo_btrack')
            else:
                self.on_iteration_start()
                self.xstep()
                self.ystep()
            self.timer.stop(['solve_wo_rsdl', 'solve_wo_btrack'])
            if not self.opt['FastSolve']:
                frcxd, adapt_tol = self.compute_residuals()
            self.timer.start('solve_wo_rsdl')
            self.timer.stop(['solve_wo_func', 'solve_wo_rsdl', 'solve_wo_btrack'])
            if not self.opt['FastSolve']:
                itst = self.iteration_stats(self.k, frcxd)
                self.itstat.append(itst)
                self.display_status(fmtstr, itst)
            self.timer.start(['solve_wo_func', 'solve_wo_rsdl', 'solve_wo_btrack'])
            if self.opt['Callback'] is not None:
                if self.opt['Callback'](self):
                    break
            if not self.opt['FastSolve']:
                if frcxd < adapt_tol:
                    break
        self.k += 1
        self.finish_solve()
        self.timer.stop(labels)
        self.display_end(nsep)
        # (solvers driven by an outer loop that never looks at the return value set
        # `_return_min = False`: for a device-resident solver it is a device-to-host copy)
        return self.getmin() if getattr(self, '_return_min', True) else None

    def finish_solve(self):
        pass

    def fused_iteration(self):
        """Run on_iteration_start + xstep + ystep as one device call when the
        problem class can; return False to have the loop compose them."""
        return False

    def getmin(self):
        return self.X

    def var_momentum(self):
        """Nesterov's rule needs the current t, the linear rules the iteration."""
        return self.t if isinstance(self.momentum, MomentumNesterov) else self.k

    def compute_residuals(self):
        r = self.rsdl()
        tol = self.opt['RelStopTol']
        if self.opt['AutoStop', 'Enabled']:
            tol = self.tau0 / (1. + self.k)
        return r, tol

    @classmethod
    def hdrtxt(cls):
        return ('Itn',) + cls.hdrtxt_objfn + ('Rsdl', 'F', 'Q', 'It_Bt', 'L')

    @classmethod
    def hdrval(cls):
        hdr = {'Itn': 'Iter'}
        hdr.update(cls.hdrval_objfun)
        hdr.update({'Rsdl': 'Rsdl', 'F': 'F_Btrack', 'Q': 'Q_Btrack',
                    'It_Bt': 'IterBTrack', 'L': 'L'})
        return hdr

    def iteration_stats(self, k, frcxd):
        tk = self.timer.elapsed(self.opt['IterTimer'])
        objfn = self.objfn if self.opt['Monotone'] else self.eval_objfn()
        tpl = (k,) + objfn + (frcxd, self.F, self.Q, self.iterBTrack, self.L) + \
            self.itstat_extra() + (tk,)
        return type(self).IterationStats(*tpl)

    def eval_objfn(self):
        fval = self.obfn_f(self.X)
        gval = self.obfn_g(self.X)
        return (fval + gval, fval, gval)

    def itstat_extra(self):
        return ()

    def getitstat(self):
        return util.transpose_ntpl_list(self.itstat)

    def display_start(self):
        if not self.opt['Verbose']:
            return '', 0
        hdrtxt = type(self).hdrtxt()
        if self.opt['Backtrack'] is None:
            hdrtxt = hdrtxt[0:-4]
        hdrstr, fmtstr, nsep = common.solve_status_str(
            hdrtxt, fmtmap={'It_Bt': '%5d'}, fwdth0=type(self).fwiter,
            fprec=type(self).fpothr)
        if self.opt['StatusHeader']:
            print(hdrstr)
            print("-" * nsep)
        return fmtstr, nsep

    def display_status(self, fmtstr, itst):
        if self.opt['Verbose']:
            hdrval = type(self).hdrval()
            row = tuple(getattr(itst, hdrval[col]) for col in type(self).hdrtxt())
            if self.opt['Backtrack'] is None:
                row = row[0:-4]
            print(fmtstr % row)

    def display_end(self, nsep):
        if self.opt['Verbose'] and self.opt['StatusHeader']:
            print("-" * nsep)

    # -- to be provided by problem classes -------------------------------------
    def on_iteration_start(self):
        raise NotImplementedError()

    def xstep(self, grad=None):
        raise NotImplementedError()

    def ystep(self):
        raise NotImplementedError()

    def grad_f(self, V=None):
        raise NotImplementedError()

    def prox_g(self, V):
        raise NotImplementedError()

    def hessian_f(self, V):
        raise NotImplementedError()

    def obfn_f(self, X=None):
        raise NotImplementedError()

    def obfn_g(self, X):
        raise NotImplementedError()

    def rsdl(self):
        raise NotImplementedError()


class PGMDFT(PGM):
    r"""PGM with iterates, gradient and momentum handled in the DFT domain on the
    GPU (the role of sporco/pgm/pgm.py:708-894).  A derived class provides
    ``self.dev`` (a :class:`sporco_amd._lib.Solver`) and the problem-specific
    ``grad_f`` / ``prox_step`` / objective pieces."""

    class Options(PGM.Options):
        defaults = copy.deepcopy(PGM.Options.defaults)

        def __init__(self, opt=None):
            PGM.Options.__init__(self, {} if opt is None else opt)

    def __init__(self, xshape, Nv, axisN, dtype, opt=None):
        if opt is None:
            opt = PGMDFT.Options()
        super(PGMDFT, self).__init__(xshape, dtype, opt)
        self.Nv = Nv
        self.axisN = axisN

    # -- handles of the device-resident iterates ---------------------------------
    # state-variable ids of the iterate family this solver works on
    _v = {'x': _lib.VAR_XF, 'y': _lib.VAR_YF, 'xprv': _lib.VAR_XFPRV,
          'yprv': _lib.VAR_YFPRV, 'g': _lib.VAR_GF, 't0': _lib.VAR_T0,
          't1': _lib.VAR_T1, 't2': _lib.VAR_T2}

    def var_x(self):
        return self._v['x']

    def var_y(self, y=None):
        if y is not None and y != self._v['y']:
            self.dev.copy(self._v['y'], y)
            self.invalidate(self._v['y'])
        return self._v['y']

    def var_xprv(self):
        return self._v['xprv']

    def scratch(self, i):
        """Device scratch array i (0..2) of the same shape as the iterates."""
        return self._v['t%d' % i]

    def invalidate(self, *variables):
        """Forget cached scalars / host copies of rewritten device arrays."""
        for v in variables:
            self._fcache.pop(v, None)
            self._cache.pop(v, None)

    # -- one proximal-gradient step ---------------------------------------------------
    def xstep(self, gradf=None):
        """Vf = Yf - gradf/L, X = prox_g(irfftn(Vf)), Xf = rfftn(X)
        (sporco/pgm/pgm.py:779-811)."""
        if gradf is None:
            gradf = self.grad_f()
        if self.stepsizepolicy is not None:
            if self.k > 1:
                self.L = self.dtype.type(self.stepsizepolicy.update(self, gradf))
            if isinstance(self.stepsizepolicy, StepSizePolicyBB):
                self.stepsizepolicy.store_prev_state(self, self.var_x(), gradf)
        self.prox_step(gradf)
        if self.opt['Monotone'] and self.k > 0:
            self.dev.copy(self._v['t2'], self._v['x'])          # ZZf = Xf.copy()
            self.objfn = self.eval_objfn()
            if self.objfn_prev[0] < self.objfn[0]:
                # objective went up: fall back to the previous iterate
                self.dev.copy(self._v['x'], self._v['xprv'])
                self.invalidate(self._v['x'])
                self.objfn = self.objfn_prev
        return gradf

    def ystep(self):
        """Yf = Xf + ((t_prev - 1)/t)(Xf - Xfprv) [+ (t_prev/t)(ZZf - Xf) when
        Monotone] (sporco/pgm/pgm.py:815-831)."""
        tprv = self.t
        self.t = self.momentum.update(self.var_momentum())
        beta = (tprv - 1.) / self.t
        if self.opt['Monotone'] and self.k > 0:
            gamma = tprv / self.t
            self.dev.lincomb(self._v['y'], 1.0 + beta - gamma, self._v['x'], -beta,
                             self._v['xprv'], gamma, self._v['t2'])
        else:
            self.dev.lincomb(self._v['y'], 1.0 + beta, self._v['x'], -beta, self._v['xprv'])
        self.invalidate(self._v['y'])

    def on_iteration_start(self):
        """Xfprv = Xf, Yfprv = Yf (sporco/pgm/pgm.py:835-846)."""
        self.dev.copy(self._v['xprv'], self._v['x'])
        self.invalidate(self._v['xprv'])
        if not self.opt['FastSolve'] or isinstance(self.backtrack, BacktrackRobust):
            self.dev.copy(self._v['yprv'], self._v['y'])
            self.invalidate(self._v['yprv'])
        if self.opt['Monotone']:
            if self.k == 0:
                self.objfn = self.eval_objfn()
            self.objfn_prev = self.objfn

    def eval_Dxy(self):
        return (self._v['x'], self._v['y'])

    def rsdl(self):
        """rfl2norm2(Xf - Yfprv) (pgm/cbpdn.py:314-320, pgm/ccmod.py:326-332)."""
        return self.dev.pair_stats(self._v['x'], self._v['yprv'])[0]

    def eval_linear_approx(self, Dxy, gradY):
        return self.dev.pair_stats(Dxy[0], Dxy[1], gradY)[1]

    def finish_solve(self):
        self.dev.sync()
