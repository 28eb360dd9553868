"""ADMM solver framework (host side).

Keeps the template-method contract of the reference's ``sporco.admm.admm``
(ADMM: sporco/admm/admm.py:28-786, ADMMEqual: :791-983): ``solve()`` drives
``xstep / relax_AX / ystep / ustep / compute_residuals / iteration_stats /
update_rho`` on ``self``, with the same Options tree, IterationStats fields,
timers and stopping rule.  What differs is where the arrays live: concrete
problem classes keep them on the GPU and override the steps with device calls;
this module only ever touches scalars.
"""

import copy

import numpy as np

from .. import cdict
from .. import common
from .. import util
from ..fft import real_dtype


STEP_HOOKS = ('xstep', 'relax_AX', 'ystep', 'ustep', 'save_yprev', 'rsdl_r', 'rsdl_s', 'rsdl_rn',
              'rsdl_sn', 'cnst_A', 'cnst_AT', 'cnst_B', 'cnst_c', 'obfn_f', 'obfn_g',
              'obfn_fvar', 'obfn_gvar', 'obfn_g0', 'obfn_g1', 'obfn_g0var', 'obfn_g1var')


def refuse_step_overrides(obj, names=STEP_HOOKS):
    """For solver classes whose whole iteration is one device call: overriding or
    monkey-patching a step method of the reference's template (admm.py:331-367) cannot take
    effect, so it raises instead of being silently ignored.  An override is a definition of
    the name anywhere below this package's classes, or on the instance."""
    cls = type(obj)
    for name in names:
        if name in obj.__dict__:
            raise NotImplementedError(
                "%s: instance attribute %r replaces a step that runs on the device as part "
                "of one fused call; it would have no effect" % (cls.__name__, name))
        for c in cls.__mro__:
            if name in c.__dict__:
                if not c.__module__.startswith('sporco_amd.'):
                    raise NotImplementedError(
                        "%s.%s overrides a step that runs on the device as part of one fused "
                        "call; it would have no effect" % (c.__name__, name))
                break


class ADMM(common.IterativeSolver):
    r"""Base class: minimise f(x) + g(y) subject to Ax + By = c."""

    class Options(cdict.ConstrainedDict):
        """ADMM options; keys and defaults as sporco/admm/admm.py:148-161."""

        defaults = {'FastSolve': False, 'Verbose': False, 'StatusHeader': True,
                    'DataType': None, 'MaxMainIter': 1000, 'IterTimer': 'solve',
                    'AbsStopTol': 0.0, 'RelStopTol': 1e-3, 'RelaxParam': 1.0,
                    'rho': None,
                    'AutoRho': {'Enabled': False, 'Period': 10, 'Scaling': 2.0,
                                'RsdlRatio': 10.0, 'RsdlTarget': None,
                                'AutoScaling': False, 'StdResiduals': False},
                    'Y0': None, 'U0': None, 'Callback': None}

        def __init__(self, opt=None):
            cdict.ConstrainedDict.__init__(self, {} if opt is None else opt)

    fwiter = 4
    fpothr = 2
    itstat_fields_objfn = ('ObjFun', 'FVal', 'GVal')
    itstat_fields_alg = ('PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')
    itstat_fields_extra = ()
    hdrtxt_objfn = ('Fnc', 'f', 'g')
    hdrval_objfun = {'Fnc': 'ObjFun', 'f': 'FVal', 'g': 'GVal'}

    def __new__(cls, *args, **kwargs):
        obj = super(ADMM, cls).__new__(cls)
        obj.timer = util.Timer(['init', 'solve', 'solve_wo_func', 'solve_wo_rsdl'])
        obj.timer.start('init')
        return obj

    def __init__(self, Nx, yshape, ushape, dtype, opt=None):
        if opt is None:
            opt = ADMM.Options()
        if not isinstance(opt, ADMM.Options):
            raise TypeError('Parameter opt must be an instance of ADMM.Options')
        self.opt = opt
        self.Nx = Nx
        self.Nc = int(np.prod(np.array(ushape)))
        self.set_dtype(opt, dtype)
        rdt = real_dtype(self.dtype)
        self.set_attr('rho', opt['rho'], dval=1.0, dtype=rdt)
        self.set_attr('rho_tau', opt['AutoRho', 'Scaling'], dval=2.0, dtype=rdt)
        self.set_attr('rho_mu', opt['AutoRho', 'RsdlRatio'], dval=10.0, dtype=rdt)
        self.set_attr('rho_xi', opt['AutoRho', 'RsdlTarget'], dval=1.0, dtype=rdt)
        self.set_attr('rlx', opt['RelaxParam'], dval=1.0, dtype=rdt)
        self.init_state(yshape, ushape)
        self.itstat = []
        self.k = 0

    # -- state initialisation (host arrays here; device classes override) --
    def init_state(self, yshape, ushape):
        if not hasattr(self, 'X'):
            self.X = None
        if self.opt['Y0'] is None:
            self.Y = self.yinit(yshape)
        else:
            self.Y = self.opt['Y0'].astype(self.dtype, copy=True)
        self.Yprev = self.Y.copy()
        if self.opt['U0'] is None:
            self.U = self.uinit(ushape)
        else:
            self.U = self.opt['U0'].astype(self.dtype, copy=True)

    def yinit(self, yshape):
        return np.zeros(yshape, dtype=self.dtype)

    def uinit(self, ushape):
        return np.zeros(ushape, dtype=self.dtype)

    # -- main loop ---------------------------------------------------------------
    def _needs_residuals(self):
        return bool(self.opt['AutoRho', 'Enabled'] or not self.opt['FastSolve'])

    def iteration(self):
        """One pass of the reference loop body up to the residuals
        (sporco/admm/admm.py:333-352); returns ``(r, s, epri, edua)`` or None."""
        self.save_yprev()
        self.xstep()
        self.relax_AX()
        self.ystep()
        self.ustep()
        self.timer.stop('solve_wo_rsdl')
        res = self.compute_residuals() if self._needs_residuals() else None
        self.timer.start('solve_wo_rsdl')
        return res

    def save_yprev(self):
        self.Yprev = self.Y.copy()

    def solve(self):
        """Run (or continue) the iterations; see sporco/admm/admm.py:293-389."""
        fmtstr, nsep = self.display_start()
        all_timers = ['solve', 'solve_wo_func', 'solve_wo_rsdl']
        self.timer.start(all_timers)
        for self.k in range(self.k, self.k + self.opt['MaxMainIter']):
            res = self.iteration()
            if res is not None:
                r, s, epri, edua = res
            self.timer.stop(['solve_wo_func', 'solve_wo_rsdl'])
            if not self.opt['FastSolve']:
                itst = self.iteration_stats(self.k, r, s, epri, edua)
                self.itstat.append(itst)
                self.display_status(fmtstr, itst)
            self.timer.start(['solve_wo_func', 'solve_wo_rsdl'])
            self.timer.stop('solve_wo_rsdl')
            if res is not None:
                self.update_rho(self.k, r, s)
            self.timer.start('solve_wo_rsdl')
            if self.opt['Callback'] is not None:
                if self.opt['Callback'](self):
                    break
            if res is not None and r < epri and s < edua:
                break
        self.k += 1
        self.finish_solve()
        self.timer.stop(all_timers)
        self.display_end(nsep)
        # (solvers driven by an outer loop that never looks at the return value set
        # `_return_min = False`: for a device-resident solver it is a device-to-host copy)
        return self.getmin() if getattr(self, '_return_min', True) else None

    def finish_solve(self):
        """Hook run before the solve timers stop (device classes synchronise)."""

    def getmin(self):
        return self.X

    # -- steps to be provided by problem classes -----------------------------------
    def xstep(self):
        raise NotImplementedError()

    def ystep(self):
        raise NotImplementedError()

    def ustep(self):
        self.U += self.rsdl_r(self.AX, self.Y)

    def relax_AX(self):
        self.AXnr = self.cnst_A(self.X)
        if self.rlx == 1.0:
            self.AX = self.AXnr
        else:
            if not hasattr(self, '_cnst_c'):
                self._cnst_c = self.cnst_c()
            alpha = self.rlx
            self.AX = alpha * self.AXnr - (1 - alpha) * (self.cnst_B(self.Y) - self._cnst_c)

    # -- residuals and stopping thresholds ------------------------------------------
    def residual_norms(self):
        """Return (||r||, ||s||/rho... ) building blocks: ``(nr, ns, nax, ny, nu)`` with
        nr = ||AXnr + By - c||, ns = ||rho A^T B (Y - Yprev)||, and the norms
        entering the normalisation terms.  Host-array default."""
        nr = np.linalg.norm(self.rsdl_r(self.AXnr, self.Y))
        ns = np.linalg.norm(self.rsdl_s(self.Yprev, self.Y))
        return nr, ns, self.rsdl_rn(self.AXnr, self.Y), self.rsdl_sn(self.U)

    def compute_residuals(self):
        """Residuals and tolerances, standard or normalised
        (sporco/admm/admm.py:462-486)."""
        nr, ns, rn, sn = self.residual_norms()
        abstol, reltol = self.opt['AbsStopTol'], self.opt['RelStopTol']
        if self.opt['AutoRho', 'StdResiduals']:
            r, s = nr, ns
            epri = np.sqrt(self.Nc) * abstol + rn * reltol
            edua = np.sqrt(self.Nx) * abstol + sn * reltol
        else:
            if rn == 0.0:
                rn = 1.0
            if sn == 0.0:
                sn = 1.0
            r, s = nr / rn, ns / sn
            epri = np.sqrt(self.Nc) * abstol / rn + reltol
            edua = np.sqrt(self.Nx) * abstol / sn + reltol
        return r, s, epri, edua

    def rsdl_r(self, AX, Y):
        if not hasattr(self, '_cnst_c'):
            self._cnst_c = self.cnst_c()
        return AX + self.cnst_B(Y) - self._cnst_c

    def rsdl_s(self, Yprev, Y):
        return self.rho * self.cnst_AT(self.cnst_B(Y - Yprev))

    def rsdl_rn(self, AX, Y):
        if not hasattr(self, '_nrm_cnst_c'):
            self._nrm_cnst_c = np.linalg.norm(self.cnst_c())
        return max(np.linalg.norm(AX), np.linalg.norm(self.cnst_B(Y)), self._nrm_cnst_c)

    def rsdl_sn(self, U):
        return self.rho * np.linalg.norm(self.cnst_AT(U))

    def cnst_A(self, X):
        raise NotImplementedError()

    def cnst_AT(self, X):
        raise NotImplementedError()

    def cnst_B(self, Y):
        raise NotImplementedError()

    def cnst_c(self):
        raise NotImplementedError()

    # -- adaptive penalty parameter ----------------------------------------------------
    def rho_scale_factor(self, k, r, s):
        """Multiplier applied to rho at iteration k (1.0 = unchanged):
        the decision logic of sporco/admm/admm.py:552-571."""
        if not self.opt['AutoRho', 'Enabled']:
            return 1.0
        if k == 0 or np.mod(k + 1, self.opt['AutoRho', 'Period']) != 0:
            return 1.0
        tau, mu, xi = self.rho_tau, self.rho_mu, self.rho_xi
        if self.opt['AutoRho', 'AutoScaling']:
            if s == 0.0 or r == 0.0:
                rhomlt = tau
            else:
                rhomlt = np.sqrt(r / (s * xi) if r > s * xi else (s * xi) / r)
                if rhomlt > tau:
                    rhomlt = tau
        else:
            rhomlt = tau
        if r > xi * mu * s:
            return rhomlt
        if s > (mu / xi) * r:
            return 1.0 / rhomlt
        return 1.0

    def update_rho(self, k, r, s):
        """rho *= rsf, U /= rsf, then ``rhochange()`` (sporco/admm/admm.py:572-575)."""
        if not self.opt['AutoRho', 'Enabled']:
            return
        if k == 0 or np.mod(k + 1, self.opt['AutoRho', 'Period']) != 0:
            return
        rsf = self.rho_scale_factor(k, r, s)
        self.rho *= real_dtype(self.dtype).type(rsf)
        self.rescale_u(rsf)
        if rsf != 1.0:
            self.rhochange()

    def rescale_u(self, rsf):
        self.U /= rsf

    def rhochange(self):
        pass

    # -- statistics and display -----------------------------------------------------------
    @classmethod
    def hdrtxt(cls):
        return ('Itn',) + cls.hdrtxt_objfn + ('r', 's', u'ρ')

    @classmethod
    def hdrval(cls):
        hdrmap = {'Itn': 'Iter'}
        hdrmap.update(cls.hdrval_objfun)
        hdrmap.update({'r': 'PrimalRsdl', 's': 'DualRsdl', u'ρ': 'Rho'})
        return hdrmap

    def iteration_stats(self, k, r, s, epri, edua):
        tk = self.timer.elapsed(self.opt['IterTimer'])
        tpl = (k,) + self.eval_objfn() + (r, s, epri, edua, self.rho) + \
            self.itstat_extra() + (tk,)
        return type(self).IterationStats(*tpl)

    def eval_objfn(self):
        fval = self.obfn_f(self.X)
        gval = self.obfn_g(self.Y)
        return (fval + gval, fval, gval)

    def itstat_extra(self):
        return ()

    def getitstat(self):
        return util.transpose_ntpl_list(self.itstat)

    def display_start(self):
        if not self.opt['Verbose']:
            return '', 0
        hdrtxt = type(self).hdrtxt()
        if not self.opt['AutoRho', 'Enabled']:
            hdrtxt = hdrtxt[0:-1]
        hdrstr, fmtstr, nsep = common.solve_status_str(
            hdrtxt, fwdth0=type(self).fwiter, fprec=type(self).fpothr)
        if self.opt['StatusHeader']:
            print(hdrstr)
            print("-" * nsep)
        return fmtstr, nsep

    def display_status(self, fmtstr, itst):
        if self.opt['Verbose']:
            hdrval = type(self).hdrval()
            row = tuple(getattr(itst, hdrval[col]) for col in type(self).hdrtxt())
            if not self.opt['AutoRho', 'Enabled']:
                row = row[0:-1]
            print(fmtstr % row)

    def display_end(self, nsep):
        if self.opt['Verbose'] and self.opt['StatusHeader']:
            print("-" * nsep)

    def var_x(self):
        return self.X

    def var_y(self):
        return self.Y

    def var_u(self):
        return self.U

    def obfn_f(self, X):
        raise NotImplementedError()

    def obfn_g(self, Y):
        raise NotImplementedError()


class ADMMEqual(ADMM):
    r"""ADMM with the constraint x = y (A = I, B = -I, c = 0)."""

    class Options(ADMM.Options):
        """Adds ``fEvalX``, ``gEvalY``, ``ReturnX`` (sporco/admm/admm.py:833-834)."""

        defaults = copy.deepcopy(ADMM.Options.defaults)
        defaults.update({'fEvalX': True, 'gEvalY': True, 'ReturnX': True})

        def __init__(self, opt=None):
            ADMM.Options.__init__(self, {} if opt is None else opt)

    def __init__(self, xshape, dtype, opt=None):
        if opt is None:
            opt = ADMMEqual.Options()
        Nx = int(np.prod(np.array(xshape)))
        super(ADMMEqual, self).__init__(Nx, xshape, xshape, dtype, opt)

    def getmin(self):
        return self.X if self.opt['ReturnX'] else self.Y

    def relax_AX(self):
        self.AXnr = self.X
        if self.rlx == 1.0:
            self.AX = self.X
        else:
            alpha = self.rlx
            self.AX = alpha * self.X + (1 - alpha) * self.Y

    def obfn_fvar(self):
        return self.X if self.opt['fEvalX'] else self.Y

    def obfn_gvar(self):
        return self.Y if self.opt['gEvalY'] else self.X

    def eval_objfn(self):
        fval = self.obfn_f(self.obfn_fvar())
        gval = self.obfn_g(self.obfn_gvar())
        return (fval + gval, fval, gval)

    def cnst_A(self, X):
        return X

    def cnst_AT(self, Y):
        return Y

    def cnst_B(self, Y):
        return -Y

    def cnst_c(self):
        return 0.0

    def rsdl_r(self, AX, Y):
        return AX - Y

    def rsdl_s(self, Yprev, Y):
        return self.rho * (Yprev - Y)

    def rsdl_rn(self, AX, Y):
        return max(np.linalg.norm(AX), np.linalg.norm(Y))

    def rsdl_sn(self, U):
        return self.rho * np.linalg.norm(U)
