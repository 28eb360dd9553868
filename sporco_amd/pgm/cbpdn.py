"""FISTA convolutional sparse coding on the GPU (PGM ConvBPDN).

Drop-in for ``sporco.pgm.cbpdn.ConvBPDN`` (sporco/pgm/cbpdn.py:29-383): same
constructor, Options, IterationStats fields (``Iter, ObjFun, DFid, RegL1,
Rsdl, F_Btrack, Q_Btrack, IterBTrack, L, Time``) and attributes.  Per
iteration the device runs

    gradient   GF = conj(Df) (sum_m Df Yf - Sf)            (one fused kernel)
    prox step  Vf = Yf - GF/L -> irfftn -> soft threshold -> rfftn
    momentum   Yf = Xf + beta (Xf - Xfprv)                  (one kernel)

with backtracking / step-size policies composed from device reductions.
"""

import copy

import numpy as np

from . import pgm
from .backtrack import BacktrackStandard, BacktrackRobust
from .stepsize import StepSizePolicyCauchy, StepSizePolicyBB
from .. import _lib
from .. import cnvrep as cr
from ..admm.cbpdn import _DeviceArray, _broadcastable, _reshaped_options

__all__ = ['ConvBPDN', 'ConvBPDNMask']


class ConvBPDN(pgm.PGMDFT):
    r"""Minimise (1/2)||sum_m d_m * x_m - s||_2^2 + lambda sum_m ||x_m||_1 by
    accelerated proximal gradient."""

    _dim1_ok = True


    class Options(pgm.PGMDFT.Options):
        """Adds ``NonNegCoef``, ``NoBndryCross``, ``L1Weight``; default ``L`` 500
        (sporco/pgm/cbpdn.py:112-118)."""

        defaults = copy.deepcopy(pgm.PGMDFT.Options.defaults)
        defaults.update({'NonNegCoef': False, 'NoBndryCross': False})
        defaults.update({'L1Weight': 1.0})
        defaults.update({'L': 500.0})

        def __init__(self, opt=None):
            pgm.PGMDFT.Options.__init__(self, {} if opt is None else opt)

    itstat_fields_objfn = ('ObjFun', 'DFid', 'RegL1')
    hdrtxt_objfn = ('Fnc', 'DFid', u'Regℓ1')
    hdrval_objfun = {'Fnc': 'ObjFun', 'DFid': 'DFid', u'Regℓ1': 'RegL1'}

    X = _DeviceArray(_lib.VAR_X)
    Xf = _DeviceArray(_lib.VAR_XF)
    Yf = _DeviceArray(_lib.VAR_YF)
    Xfprv = _DeviceArray(_lib.VAR_XFPRV)
    Yfprv = _DeviceArray(_lib.VAR_YFPRV)
    Vf = _DeviceArray(_lib.VAR_VF)
    Df = _DeviceArray(_lib.VAR_DF)
    Sf = _DeviceArray(_lib.VAR_SF)

    def __init__(self, D, S, lmbda=None, opt=None, dimK=None, dimN=2, device=0, stream=None,
                 reducer=None):
        """``D, S, lmbda, opt, dimK, dimN`` as in the reference (pgm/cbpdn.py:147-218).  Backend
        keywords: ``device``, ``stream``, and ``reducer`` (:class:`sporco_amd.dist.TorchReducer`)
        when ``S`` is this rank's block of the images: every sum the iteration takes over the
        coefficient arrays is then all-reduced (:class:`sporco_amd.dist.ReducingSolver`), so all
        ranks follow the single-process step-size, backtracking and stopping decisions."""
        self._reducer = reducer
        if opt is None:
            opt = ConvBPDN.Options()
        # dimN = 1 (signals): the two-dimensional machinery on arrays with a unit first axis, the
        # public arrays in the reference's dimN = 1 shapes (see admm.cbpdn.GenericConvBPDN)
        self._dim1 = False
        if dimN == 1 and type(self)._dim1_ok:
            self._dim1 = True
            D, S, dimN = np.asarray(D)[np.newaxis], np.asarray(S)[np.newaxis], 2
            opt = _reshaped_options(opt, ('L1Weight',), lambda a: a[np.newaxis])
        # dimN = 3 (volumes): the first two axes folded, on a volume handle (admm.cbpdn.GenericConvBPDN)
        self._dim3 = None
        if dimN == 3 and type(self)._dim1_ok:
            if opt['NoBndryCross'] or reducer is not None:
                raise NotImplementedError("dimN = 3: no NoBndryCross, no image shards")
            self._dim3, D, S = cr.volume_problem(D, S, dimK)
            dimK, dimN = 1, 2
            opt = _reshaped_options(opt, ('L1Weight',), lambda a: cr.fold3(a, *self._dim3))
        if dimN != 2:
            raise NotImplementedError("sporco_amd handles dimN = 2 (images) and, for ConvBPDN, "
                                      "dimN = 1 (signals) and 3 (volumes)")
        if not (np.isrealobj(D) and np.isrealobj(S)):
            raise NotImplementedError("sporco_amd handles real-valued D and S")
        if not hasattr(self, 'cri'):
            self.cri = cr.CSC_ConvRepIndexing(D, S, dimK=dimK, dimN=dimN)
        self.set_dtype(opt, S.dtype)
        if self.dtype not in (np.float32, np.float64):
            raise TypeError("sporco_amd works in float32 or float64, not %s" % self.dtype)
        self._device, self._stream = device, stream
        self._new_handle()
        self.D = np.asarray(D.reshape(self.cri.shpD), dtype=self.dtype)
        self.S = np.asarray(S.reshape(self.cri.shpS), dtype=self.dtype)
        self.dev.set_signal(self.S)
        self.setdict()
        if lmbda is None:
            lmbda = 0.1 * self.dev.dhs_absmax()            # pgm/cbpdn.py:209-214
        self.lmbda = self.dtype.type(lmbda)
        self.wl1 = np.asarray(opt['L1Weight'], dtype=self.dtype)
        self._upload_weights()
        super(ConvBPDN, self).__init__(self.cri.shpX, self.cri.Nv, self.cri.axisN, S.dtype, opt)

    # -- device plumbing ----------------------------------------------------------
    def _new_handle(self):
        H, W = self.cri.Nv
        self.dev = _lib.Solver(H, W, self.cri.C, self.cri.K, self.cri.M, self.dtype,
                               device=self._device, stream=self._stream, Cd=self.cri.Cd,
                               depth=self._dim3[0] if getattr(self, '_dim3', None) else 1)
        if getattr(self, '_reducer', None) is not None:
            from ..dist import ReducingSolver
            self.dev = ReducingSolver(self.dev, self._reducer)
        self._cache = {}
        self._fcache = {}
        self._rl1 = 0.0
        self._wl1_scalar = 1.0

    def _fetch(self, var):
        if var not in self._cache:
            a = self.dev.download(var)
            if getattr(self, '_dim1', False) and a.ndim >= 2 and a.shape[0] == 1:
                a = a[0]
            if getattr(self, '_dim3', None):
                a = cr.unfold3(a, *self._dim3)
            self._cache[var] = a
        return self._cache[var]

    def _store(self, var, value):
        if value is None:
            return
        value = np.asarray(value)
        if getattr(self, '_dim1', False) and value.ndim == 4:
            value = value[np.newaxis]
        if getattr(self, '_dim3', None) and value.ndim == 6:
            value = cr.fold3(value, *self._dim3)
        self.dev.upload(var, value)
        self.invalidate(var)

    def _upload_weights(self):
        w = self.wl1
        if w.size == 1:
            self._wl1_scalar = float(w.ravel()[0])
            self.dev.set_l1_weight(None)
        else:
            self._wl1_scalar = 1.0
            w5 = w.reshape(cr.l1Wshape(w, self.cri)) if w.ndim != 5 else w
            self.dev.set_l1_weight(_broadcastable(w5, self.cri.shpX))

    def init_state(self, xshape):
        """X = X0 or 0; Xf = rfftn(X); Yf = Xf (pgm/cbpdn.py:222-233)."""
        if self.opt['X0'] is not None:
            self.X = np.asarray(self.opt['X0']).astype(self.dtype, copy=True)
            self.dev.fft_var(_lib.VAR_X, _lib.VAR_XF)
            self.dev.copy(_lib.VAR_YF, _lib.VAR_XF)
        self.Y = None

    def __getstate__(self):
        state = self.__dict__.copy()
        for key in ('dev', '_cache', '_fcache'):
            state.pop(key, None)
        state['_saved_arrays'] = {v: self.dev.download(v) for v in
                                  (_lib.VAR_X, _lib.VAR_XF, _lib.VAR_YF, _lib.VAR_XFPRV,
                                   _lib.VAR_YFPRV)}
        state['_stream'] = None
        state['_reducer'] = None          # a process group does not pickle
        return state

    def __setstate__(self, state):
        saved = state.pop('_saved_arrays')
        rl1 = state.get('_rl1', 0.0)
        self.__dict__.update(state)
        self._new_handle()
        self.dev.set_signal(self.S)
        self.setdict()
        self._upload_weights()
        for v, a in saved.items():
            self._store(v, a)
        self._rl1 = rl1

    # -- dictionary / coefficients ------------------------------------------------------
    def setdict(self, D=None):
        if D is not None:
            D = np.asarray(D, dtype=self.dtype)
            if getattr(self, '_dim1', False) and D.ndim == len(self.cri.shpD) - 1:
                D = D[np.newaxis]
            if getattr(self, '_dim3', None) and D.shape[0] != self.cri.Nv[0]:
                Dz, Hs = self._dim3
                D = D.reshape(D.shape[0:3] + (1, 1, D.shape[-1]))
                D = cr.fold3(cr.zpad(D, (Dz, Hs, self.cri.Nv[1])), Dz, Hs)
            self.D = D
        self.dev.set_dict(self.D)
        self._cache.pop(_lib.VAR_DF, None)
        self._fcache.clear()

    def getcoef(self):
        return self.X

    def _flags(self):
        f = 0
        if self.opt['NonNegCoef']:
            f |= _lib.FLAG_NONNEG
        if self.opt['NoBndryCross']:
            f |= _lib.FLAG_NOBNDRY
        return f

    def _iter_flags(self):
        """Flags of the fused iteration (the masked subclass adds the data mask)."""
        return self._flags()

    # -- whole iteration in three launches when nothing is customised -----------------
    _hook_names = ('on_iteration_start', 'xstep', 'ystep', 'grad_f', 'prox_step', 'rsdl',
                   'compute_residuals', 'eval_objfn', 'obfn_dfd', 'obfn_reg', 'obfn_f',
                   'eval_linear_approx', 'hess_quad')

    def _fused_ok(self):
        # (BacktrackStandard / BacktrackRobust themselves, not subclasses: a subclass may change
        # the search)
        bt = self.opt['Backtrack']
        pol = self.stepsizepolicy
        if (bt is not None and type(bt) not in (BacktrackStandard, BacktrackRobust)) \
                or (pol is not None and type(pol) not in (StepSizePolicyCauchy, StepSizePolicyBB)) \
                or (self.opt['Monotone'] and (bt is not None or pol is not None)) \
                or (bt is not None and pol is not None) \
                or not self.dev.uses_fused_pgm():
            # (a backtracking rule AND a step-size policy: the reference's backtrack.update calls
            # xstep() on every trial, which re-applies the policy and BB's store_prev_state each
            # time, backtrack.py:83-117 / pgm.py:786-803 -- the staged composition does exactly that)
            return False
        if pol is not None and (self.cri.Cd > 1 or self.cri.M % 2):
            return False     # (the residual slots: single-channel dictionary, unpadded filter axis)
        for name in self._hook_names:
            if name in self.__dict__ or getattr(type(self), name) is not getattr(ConvBPDN, name):
                return False
        return True

    def fused_iteration(self):
        """on_iteration_start, xstep and ystep (pgm.py:835-846, :779-831) as one
        device call (csc_pgm.h); the sums it returns serve rsdl and eval_objfn.

        With BacktrackStandard (backtrack.py:50-117) the call is a trial: F = f(X) and the
        terms of Q_L come back with it, the iterates are adopted (pgm_commit) once F <= Q, and a
        failed trial is repeated from the same Y with L gamma_u (Y = X + beta (X - Xprv) does
        not depend on L, so the momentum step rides along with the accepted trial)."""
        if not self._fused_ok():
            self._fused_sums = None
            return False
        bt = self.opt['Backtrack']
        lm = float(self.lmbda) * self._wl1_scalar
        if type(bt) is BacktrackRobust:
            return self._fused_robust(bt, lm)
        if self.opt['Monotone']:
            return self._fused_monotone(lm)
        if self.stepsizepolicy is not None and not self._fused_stepsize():
            self._fused_sums = None
            return False
        tprv = self.t
        self.t = self.momentum.update(self.var_momentum())
        beta = (tprv - 1.) / self.t
        stats = not self.opt['FastSolve']
        if bt is None:
            out = self.dev.pgm_iter(self.L, lm, beta, self._iter_flags(), self.D.shape[0],
                                    self.D.shape[1], stats)
        else:
            it = 0
            while True:
                out = self.dev.pgm_iter(self.L, lm, beta, self._iter_flags(), self.D.shape[0],
                                        self.D.shape[1], True, hold=True)
                f = out[_lib.PGM_F]
                Q = out[_lib.PGM_FY] + out[_lib.PGM_LIN] + (self.L / 2.) * out[_lib.PGM_DXY2]
                it += 1
                if f <= Q or it >= bt.maxiter:
                    if f > Q:
                        self.L *= bt.gamma_u
                    break
                self.L *= bt.gamma_u
            self.dev.pgm_commit()
            self.F, self.Q, self.iterBTrack = f, Q, it
            stats = True
        self._fused_sums = out
        self._rl1 = abs(self._wl1_scalar) * out[_lib.PGM_L1]
        self._cache.clear()
        self._fcache.clear()
        self._fcache[_lib.VAR_YFPRV] = out[_lib.PGM_FY]
        if stats:
            self._fcache[_lib.VAR_XF] = out[_lib.PGM_F]
        return True

    def _fused_stepsize(self):
        """StepSizePolicyCauchy / StepSizePolicyBB (stepsize.py:67-145) beside the fused iteration.
        The gradient at v is conj(Df) e(v) with e(v) = sum_m Df v - Sf, signal sized: one read
        pass over Yf (and, for BB, over Xf) leaves e in a residual slot, and
            <g, g> = sum G |e|^2,  <g, hessian_f g> = sum G^2 |e|^2,  G = sum_m |Df|^2,
            <dx, dg> = sum Re(conj(e(x) - e(xprv)) (e(y) - e(yprv)))
        are sums over those (sporco_amd_csc_pgm_resid / _pgm_resid_stats): no X-sized gradient
        array, 10 (Cauchy) / 11 (BB) passes an iteration instead of the staged composition's ~25.
        L changes from the third iteration on (pgm.py:790-792); BB records its two-point state
        from the first.  Returns False when BB's previous point is not in the slots (a restored
        object): that iteration is composed from the staged calls."""
        pol, dev = self.stepsizepolicy, self.dev
        bb = type(pol) is StepSizePolicyBB
        if not bb:
            if self.k > 1:
                dev.pgm_resid(_lib.VAR_YF, 0)
                s = dev.pgm_resid_stats(0)
                # (numpy division as in stepsize.py:88-90: a zero gradient gives nan / inf and a
                # warning there, not an exception)
                with np.errstate(divide='ignore', invalid='ignore'):
                    self.L = self.dtype.type(np.float64(s[1]) / np.float64(s[0]))
            return True
        par = getattr(pol, '_slot_parity', 0)
        have = getattr(pol, '_slots_filled', False)
        if self.k > 1 and not have:
            return False
        ycur, xcur, yprv, xprv = (0, 1, 2, 3) if par == 0 else (2, 3, 0, 1)
        dev.pgm_resid(_lib.VAR_YF, ycur)
        dev.pgm_resid(_lib.VAR_XF, xcur)
        if self.k > 1:
            s = dev.pgm_resid_stats(ycur, yprv, xcur, xprv)
            with np.errstate(divide='ignore', invalid='ignore'):
                L = np.float64(s[0]) / np.float64(s[2])     # (stepsize.py:139-141)
            if L < 0.:
                L = self.L
            self.L = self.dtype.type(L)
        pol._slot_parity, pol._slots_filled = par ^ 1, True
        pol.have_prev = False     # (the staged form's scratch arrays are not kept up to date)
        return True

    def _fused_monotone(self, lm):
        """One iteration of monotone FISTA (pgm.py:804-811, :826-829) on the fused kernels: a held
        trial, the objective at the new X from its sums, the commit -- and, when the objective
        went up, the reference's fall-back to the previous iterate composed from the staged
        calls (Xf = Xfprv, Yf = Xf + (t_prev/t)(ZZf - Xf); rare, and the next iteration re-enters
        the fused regime).  An accepted step IS the standard one: ZZf = Xf.  The first iteration
        evaluates the objective at the initial state (pgm.py:843-846): left to the staged loop."""
        if self.k == 0:
            self._fused_sums = None
            return False
        self.objfn_prev = self.objfn
        tprv = self.t
        self.t = self.momentum.update(self.var_momentum())
        beta = (tprv - 1.) / self.t
        dev = self.dev
        out = dev.pgm_iter(self.L, lm, beta, self._iter_flags(), self.D.shape[0], self.D.shape[1],
                           True, hold=True)
        self._fused_sums = out
        self._rl1 = abs(self._wl1_scalar) * out[_lib.PGM_L1]
        self._cache.clear()
        self._fcache.clear()
        self._fcache[_lib.VAR_YFPRV] = out[_lib.PGM_FY]
        self._fcache[_lib.VAR_XF] = out[_lib.PGM_F]
        self.objfn = self.eval_objfn()
        dev.pgm_commit()
        if self.objfn_prev[0] < self.objfn[0]:
            v = self._v
            dev.copy(v['t2'], v['x'])                 # ZZf = Xf.copy()
            dev.copy(v['x'], v['xprv'])               # Xf = Xfprv
            gamma = tprv / self.t
            dev.lincomb(v['y'], 1.0 + beta - gamma, v['x'], -beta, v['xprv'], gamma, v['t2'])
            self.objfn = self.objfn_prev
            self._fused_sums = None                   # (rsdl: Xf - Yfprv of the restored iterate)
            self._cache.clear()
            self._fcache.clear()
        return True

    def _fused_robust(self, bt, lm):
        """One iteration under BacktrackRobust (backtrack.py:162-208) on the fused kernels:
        y = (Tk xprv + t Z) / T is formed on the device in the layout the iterates are in
        (lincomb), the trial is the held call of the standard rule with no momentum output
        (hold = 2), Z += t L (x - y) follows the commit.  xprv of the reference is X at the
        start of the iteration (pgm.py:835-846), i.e. the device's current Xf; the residual
        compares the new X with the Y the iteration STARTED from (pgm/cbpdn.py:314-320),
        which after the commit is what VAR_YF holds."""
        dev = self.dev
        Z = self.scratch(2)
        if not bt.have_z:
            dev.copy(Z, _lib.VAR_XF)
            dev.copy(_lib.VAR_YFPRV, _lib.VAR_YF)      # (on_iteration_start of the first iteration)
            bt.have_z = True
        self.L *= bt.gamma_d
        it = 0
        while True:
            t = float(1. + np.sqrt(1. + 4. * self.L * bt.Tk)) / (2. * self.L)
            T = bt.Tk + t
            dev.lincomb(_lib.VAR_YF, bt.Tk / T, _lib.VAR_XF, t / T, Z)
            out = dev.pgm_iter(self.L, lm, 0.0, self._iter_flags(), self.D.shape[0],
                               self.D.shape[1], True, hold=2)
            f = out[_lib.PGM_F]
            Q = out[_lib.PGM_FY] + out[_lib.PGM_LIN] + (self.L / 2.) * out[_lib.PGM_DXY2]
            it += 1
            if f <= Q or it >= bt.maxiter:
                if f > Q:
                    self.L *= bt.gamma_u
                break
            self.L *= bt.gamma_u
        dev.pgm_commit()
        bt.Tk = T
        tl = t * float(self.L)
        dev.lincomb(Z, 1.0, Z, tl, _lib.VAR_XF, -tl, _lib.VAR_YFPRV)
        self.F, self.Q, self.iterBTrack = f, Q, it
        if not self.opt['FastSolve']:
            out = list(out)
            out[_lib.PGM_RSDL] = dev.pair_stats(_lib.VAR_XF, _lib.VAR_YF)[0]
        self._fused_sums = out
        self._rl1 = abs(self._wl1_scalar) * out[_lib.PGM_L1]
        self._cache.clear()
        self._fcache.clear()
        self._fcache[_lib.VAR_XF] = out[_lib.PGM_F]
        return True

    def rsdl(self):
        if getattr(self, '_fused_sums', None) is not None:
            return self._fused_sums[_lib.PGM_RSDL]
        return super(ConvBPDN, self).rsdl()

    # -- smooth term ---------------------------------------------------------------------
    def grad_f(self, V=None):
        """GF = conj(Df)(sum_m Df V - Sf) at V (default Yf); returns the handle of
        the gradient array (pgm/cbpdn.py:263-279).  f(V) comes back for free."""
        if V is None:
            V = _lib.VAR_YF
        out = self.dev.pgm_grad(V)
        self._fcache[V] = out[_lib.PGM_F]
        self.invalidate(_lib.VAR_GF)
        return _lib.VAR_GF

    def obfn_f(self, Xf=None):
        """(1/2)||sum_m Df Xf - Sf||^2 in the unnormalised DFT domain
        (pgm/cbpdn.py:358-372)."""
        if Xf is None:
            Xf = _lib.VAR_XF
        if Xf not in self._fcache:
            self._fcache[Xf] = self.dev.pgm_eval(Xf)[_lib.PGM_F]
        return self._fcache[Xf]

    def prox_step(self, gradf):
        if gradf != _lib.VAR_GF:
            self.dev.copy(_lib.VAR_GF, gradf)
        out = self.dev.pgm_prox_step(self.L, float(self.lmbda) * self._wl1_scalar, self._flags(),
                                     self.D.shape[0], self.D.shape[1])
        self._rl1 = abs(self._wl1_scalar) * out[_lib.PGM_L1]
        self.invalidate(_lib.VAR_X, _lib.VAR_XF, _lib.VAR_VF)

    def hess_quad(self, V):
        """<V, hessian_f(V)> = sum |sum_m Df V|^2 (pgm/cbpdn.py:302-312)."""
        return self.dev.pgm_eval(V)[_lib.PGM_HESS]

    # -- objective ---------------------------------------------------------------------------
    def eval_objfn(self):
        dfd = self.obfn_dfd()
        reg = self.obfn_reg()
        return (dfd + reg[0], dfd) + reg[1:]

    def obfn_dfd(self):
        if getattr(self, '_fused_sums', None) is not None:
            return self._fused_sums[_lib.PGM_DFID] / 2.0
        return self.dev.pgm_eval(_lib.VAR_XF)[_lib.PGM_DFID] / 2.0

    def obfn_reg(self):
        return (self.lmbda * self._rl1, self._rl1)

    def reconstruct(self, X=None):
        if X is None:
            var = _lib.VAR_X
        else:
            X = np.asarray(X, dtype=self.dtype)
            if getattr(self, '_dim1', False) and X.ndim == 4:
                X = X[np.newaxis]
            if getattr(self, '_dim3', None) and X.ndim == 6:
                X = cr.fold3(X, *self._dim3)
            self.dev.upload(_lib.VAR_AX, X)
            var = _lib.VAR_AX
        r = self.dev.reconstruct(var)[..., 0]
        if getattr(self, '_dim3', None):
            return cr.unfold3(r, *self._dim3)
        return r[0] if getattr(self, '_dim1', False) else r


class ConvBPDNMask(ConvBPDN):
    r"""FISTA for convolutional BPDN with a spatial mask in the data fidelity term,
    (1/2) ||W (sum_m d_m * x_m - s)||_2^2 + lambda sum_m ||x_m||_1 (reference class:
    sporco/pgm/cbpdn.py:387-506).  The gradient takes the residual to the spatial domain,
    weights it by W^2 and brings it back (``sporco_amd_csc_masked_grad``); everything else is
    the unmasked solver."""

    _dim1_ok = False


    def __init__(self, D, S, lmbda, W=None, opt=None, dimK=None, dimN=2, **backend):
        super(ConvBPDNMask, self).__init__(D, S, lmbda, opt, dimK=dimK, dimN=dimN, **backend)
        if W is None:
            W = np.array([1.0], dtype=self.dtype)
        W = np.asarray(W)
        shp = (1,) * 5 if W.size == 1 else cr.mskWshape(W, self.cri)
        self.W = np.asarray(W.reshape(shp), dtype=self.dtype)
        self._upload_mask()

    def _upload_mask(self):
        H, Wd = self.cri.Nv
        self.dev.set_data_mask(_broadcastable(self.W, (H, Wd, self.cri.C, self.cri.K, 1)))

    def _upload_weights(self):
        super(ConvBPDNMask, self)._upload_weights()
        if hasattr(self, 'W'):
            self._upload_mask()

    # -- the whole iteration on the fused kernels (round 4) -----------------------------------
    # Without backtracking, step-size policy or monotone restart the masked iteration is the
    # unmasked one with the residual taken through the spatial domain between the inner product
    # and the gradient (sporco_amd_csc_pgm_iter with SPORCO_AMD_FLAG_DMASK): 11 passes over an
    # X-sized spectrum instead of the staged composition's generic FFT chain.
    def _fused_ok(self):
        if self.opt['Backtrack'] is not None or self.stepsizepolicy is not None \
                or self.opt['Monotone'] or not self.dev.uses_fused_pgm() \
                or self.cri.M > 64 or self.cri.Cd > 1:
            return False
        for name in self._hook_names:
            if name in self.__dict__ or getattr(type(self), name) is not getattr(ConvBPDNMask, name):
                return False
        return True

    def _iter_flags(self):
        return self._flags() | _lib.FLAG_DMASK

    def fused_iteration(self):
        ok = super(ConvBPDNMask, self).fused_iteration()
        if ok:
            # (f in the DFT scaling is not evaluated by the masked iteration, at Y or at X: only a
            # backtracking rule reads it, and obfn_f computes it on demand)
            self._fcache.pop(_lib.VAR_YFPRV, None)
            self._fcache.pop(_lib.VAR_XF, None)
        return ok

    def grad_f(self, V=None):
        """conj(Df) rfftn(W^2 irfftn(sum_m Df V - Sf)) (pgm/cbpdn.py:454-477)."""
        if V is None:
            V = _lib.VAR_YF
        self.dev.masked_grad(V, False, True)
        self.invalidate(_lib.VAR_GF)
        return _lib.VAR_GF

    def obfn_dfd(self):
        """(1/2) ||W irfftn(sum_m Df Xf - Sf)||^2 (pgm/cbpdn.py:481-489)."""
        if getattr(self, '_fused_sums', None) is not None:
            return self._fused_sums[_lib.PGM_DFID] / 2.0
        return self.dev.masked_grad(_lib.VAR_XF, False, False)[_lib.PGM_DFID] / 2.0

    def obfn_f(self, Xf=None):
        """(1/2) ||rfftn(W irfftn(sum_m Df Xf - Sf))||^2, DFT scaling kept
        (pgm/cbpdn.py:493-506)."""
        if Xf is None:
            Xf = _lib.VAR_XF
        if Xf in self._fcache:
            return self._fcache[Xf]
        return self.dev.masked_grad(Xf, False, False)[_lib.PGM_F]
