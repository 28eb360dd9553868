"""ADMM convolutional sparse coding on the GPU: ConvBPDN and ConvBPDNJoint.

Drop-in for the first three classes of the reference's ``sporco.admm.cbpdn``
(GenericConvBPDN sporco/admm/cbpdn.py:30-380, ConvBPDN :386-630,
ConvBPDNJoint :636-807): same constructor signatures, Options trees,
IterationStats fields, attributes (``X, Y, U, Xf, Df, Sf, D, S, rho, lmbda,
cri, itstat, timer``) and overridable step methods.

All X-sized arrays live in HBM inside a :class:`sporco_amd._lib.Solver`
handle.  An iteration is one call into the C ABI that runs the HIP kernels

    rfftn(Y - U)  ->  Sherman-Morrison solve  ->  irfftn
    ->  relax + soft-threshold (+ l2,1 shrink) + dual update + all reductions

and hands back the sums of squares from which the host forms the residuals,
tolerances, objective and the next rho exactly as the reference does.  The
attribute accessors download on demand (``b.Y`` is a NumPy array, as in the
reference).  If a step method is overridden or monkey-patched, ``solve``
falls back to the reference's step-by-step sequence with one device call per
step, so hooks keep working.
"""

import copy

import numpy as np

from . import admm
from .. import _lib
from .. import cnvrep as cr
from ..fft import complex_dtype, real_dtype

__all__ = ['GenericConvBPDN', 'ConvBPDN', 'ConvBPDNJoint', 'ConvBPDNGradReg',
           'ConvBPDNMaskDcpl', 'AddMaskSim']


class _DeviceArray(object):
    """Attribute backed by a device-resident state array of the solver handle."""

    def __init__(self, var):
        self.var = var

    def __get__(self, obj, cls):
        if obj is None:
            return self
        return obj._fetch(self.var)

    def __set__(self, obj, value):
        obj._store(self.var, value)



def _reshaped_options(opt, keys, reshape, when=None):
    """The array-valued entries ``keys`` of ``opt`` passed through ``reshape`` -- in a private
    copy: the caller's Options object is never written to (the reference does not modify it
    either, and one Options object reused over a lambda sweep is the normal usage), and
    without an array-valued entry the object itself comes back.  ``when(array)`` restricts the
    entries that are reshaped (default: every array)."""
    todo = [k for k in keys if k in opt and opt[k] is not None and np.ndim(opt[k]) > 0 and
            (when is None or when(np.asarray(opt[k])))]
    if not todo:
        return opt
    opt = copy.deepcopy(opt)
    for k in todo:
        opt[k] = reshape(np.asarray(opt[k]))
    return opt


class GenericConvBPDN(admm.ADMMEqual):
    r"""Base class: data fidelity (1/2)||sum_m d_m * x_m - s||^2 plus a
    regulariser supplied by the derived class through ``ystep``/``obfn_reg``."""

    class Options(admm.ADMMEqual.Options):
        """Options of sporco/admm/cbpdn.py:119-164 (``AuxVarObj`` couples
        ``fEvalX``/``gEvalY``).  ``HighMemSolve`` is accepted and has no effect:
        the Sherman-Morrison denominator is always formed in-kernel with the
        current rho."""

        defaults = copy.deepcopy(admm.ADMMEqual.Options.defaults)
        defaults.update({'AuxVarObj': False, 'fEvalX': True, 'gEvalY': False,
                         'ReturnX': False, 'HighMemSolve': False,
                         'LinSolveCheck': False, 'RelaxParam': 1.8,
                         'NonNegCoef': False, 'NoBndryCross': False})
        defaults['AutoRho'].update({'Enabled': True, 'Period': 1,
                                    'AutoScaling': True, 'Scaling': 1000.0,
                                    'RsdlRatio': 1.2})

        def __init__(self, opt=None):
            admm.ADMMEqual.Options.__init__(self, {} if opt is None else opt)

        def __setitem__(self, key, value):
            admm.ADMMEqual.Options.__setitem__(self, key, value)
            if key == 'AuxVarObj':
                self['fEvalX'] = value is not True
                self['gEvalY'] = value is True

    itstat_fields_objfn = ('ObjFun', 'DFid', 'Reg')
    itstat_fields_extra = ('XSlvRelRes',)
    hdrtxt_objfn = ('Fnc', 'DFid', 'Reg')
    hdrval_objfun = {'Fnc': 'ObjFun', 'DFid': 'DFid', 'Reg': 'Reg'}

    # device-resident arrays, fetched as NumPy arrays on access
    Y = _DeviceArray(_lib.VAR_Y)
    U = _DeviceArray(_lib.VAR_U)
    X = _DeviceArray(_lib.VAR_X)
    Xf = _DeviceArray(_lib.VAR_XF)
    Df = _DeviceArray(_lib.VAR_DF)
    Sf = _DeviceArray(_lib.VAR_SF)
    Yprev = _DeviceArray(_lib.VAR_YPREV)
    AX = _DeviceArray(_lib.VAR_AX)

    # step methods whose identity decides between the fused and the staged path
    _hook_names = ('xstep', 'relax_AX', 'ystep', 'ustep', 'save_yprev',
                   'compute_residuals', 'residual_norms', 'eval_objfn', 'obfn_dfd',
                   'obfn_reg', 'iteration_stats', 'itstat_extra', 'rescale_u')
    _fused_base = None   # set after each fused-capable class definition
    # methods whose arithmetic runs inside the device kernels on EVERY path (fused or staged):
    # the objective terms and the residual definitions.  Overriding them cannot take effect
    # here, so it is refused rather than silently ignored (the reference's own AddMaskSim
    # patches obfn_gvar: that case is the FLAG_AMS treatment of this backend)
    _device_only_names = ('obfn_gvar', 'obfn_fvar', 'obfn_f', 'obfn_g', 'rsdl_r', 'rsdl_s',
                          'rsdl_rn', 'rsdl_sn', 'cnst_A', 'cnst_AT', 'cnst_B', 'cnst_c')
    # multi-channel dictionaries (cri.Cd > 1): X-step by iterated Sherman-Morrison
    # (linalg.solvemdbi_ism, cbpdn.py:277-279); the variants below that need a different
    # system matrix switch this off
    _multichannel_dict_ok = True
    # dimN = 1 through a unit first axis: the classes whose regulariser has no spatial structure
    _dim1_ok = False

    def __init__(self, D, S, opt=None, dimK=None, dimN=2, device=0, stream=None,
                 reducer=None, resident=False):
        """``D``, ``S``, ``opt``, ``dimK``, ``dimN`` as in the reference
        (sporco/admm/cbpdn.py:175-204).  Backend-only keyword arguments:
        ``device`` (HIP device index), ``stream`` (a hipStream_t to share, e.g.
        ``torch.cuda.current_stream().cuda_stream``), ``reducer`` (sums the
        per-iteration scalars over image shards held by other ranks, see
        :mod:`sporco_amd.dist`) and ``resident`` (results stay in HBM: ``solve()`` /
        ``getcoef()`` return a :class:`sporco_amd.device.DeviceArray` view instead of
        downloading).  ``S`` may itself be a ``DeviceArray`` -- e.g. the highpass output of
        :func:`sporco_amd.signal.tikhonov_filter` -- and is then taken without a host copy."""
        from ..device import DeviceArray
        self._S_dev = S if isinstance(S, DeviceArray) else None
        self._resident = bool(resident)
        if opt is None:
            opt = GenericConvBPDN.Options()
        # dimN = 1 (signals, sporco/cnvrep.py:33-198): the two-dimensional machinery on arrays with a
        # unit first axis -- a length-1 transform is the identity, so every sum and transform is
        # the one-dimensional one; the public arrays (X / Y / U, solve(), getcoef(), reconstruct())
        # keep the reference's dimN = 1 shapes (the unit axis is dropped / added at the boundary).
        self._dim1 = False
        if dimN == 1 and self._dim1_ok and self._S_dev is None:
            self._dim1 = True
            D, S, dimN = np.asarray(D)[np.newaxis], np.asarray(S)[np.newaxis], 2
            opt = _reshaped_options(opt, ('L1Weight', 'L21Weight', 'Y0', 'U0'),
                                    lambda a: a[np.newaxis])
        # dimN = 3 (volumes; the reference's examples/scripts/cdl/cbpdndl_video.py:74 is the use): the
        # first two axes folded into one -- (depth, height, W, ...) IS (depth * height, W, ...) in
        # memory -- on a handle that knows where the folded axis splits and runs the transform along
        # it as two passes (include/sporco_amd.h sporco_amd_csc_create_volume); the dictionary is
        # handed over zero-padded to the volume.  X / Y / U, solve(), getcoef(), reconstruct() keep
        # the reference's six-axis shapes; ``D`` is the folded, zero-padded array.
        self._dim3 = None
        if dimN == 3 and self._dim1_ok and self._S_dev is None:
            if opt['NoBndryCross']:
                raise NotImplementedError("dimN = 3: NoBndryCross is not offered")
            self._dim3, D, S = cr.volume_problem(D, S, dimK)
            dimK, dimN = 1, 2
            opt = _reshaped_options(opt, ('L1Weight', 'L21Weight', 'Y0', 'U0'), self._fold)
        if dimN != 2:
            raise NotImplementedError("sporco_amd handles dimN = 2 (images) and, for ConvBPDN / "
                                      "ConvBPDNJoint, dimN = 1 (signals) and 3 (volumes)")
        if not (np.isrealobj(D) and (self._S_dev is not None or np.isrealobj(S))):
            raise NotImplementedError("sporco_amd handles real-valued D and S")
        self.real_dtype = True
        if not hasattr(self, 'cri'):
            self.cri = cr.CSC_ConvRepIndexing(D, S, dimK=dimK, dimN=dimN)
        if self.cri.Cd > 1 and not self._multichannel_dict_ok:
            raise NotImplementedError(
                "%s with a multi-channel dictionary is not part of the sporco_amd hot path "
                "(ConvBPDN is: linalg.solvemdbi_ism X-step)" % type(self).__name__)
        self.set_dtype(opt, S.dtype)
        if self.dtype not in (np.float32, np.float64):
            raise TypeError("sporco_amd works in float32 or float64, not %s" % self.dtype)
        self._device, self._stream, self._reducer = device, stream, reducer
        self._new_handle()
        super(GenericConvBPDN, self).__init__(self.cri.shpX, S.dtype, opt)
        if reducer is not None:
            # residual tolerances refer to the global problem size (shards need not be equal)
            from ..dist import global_count
            self.Nx = global_count(reducer, self.Nx)
            self.Nc = global_count(reducer, self.Nc)
        self.D = np.asarray(D.reshape(self.cri.shpD), dtype=self.dtype)
        if self._S_dev is not None:
            if self._S_dev.dtype != self.dtype:
                raise TypeError("a device-resident signal must already have the solver's dtype")
            self._S_host = None            # (downloaded on first access of .S)
            self._dev.set_signal_dev(self._S_dev.ptr)
        else:
            self.S = np.asarray(S.reshape(self.cri.shpS), dtype=self.dtype)
            self._dev.set_signal(self.S)
        self.setdict()

    def _fold(self, a):
        """(depth, height, ...) -> (depth * height, ...) of a dimN = 3 array; arrays that broadcast
        along both axes lose one of the two unit axes."""
        return cr.fold3(a, *self._dim3)

    def _unfold(self, a):
        return cr.unfold3(a, *self._dim3)

    @property
    def S(self):
        """The signal in the internal layout; for a device-resident input, a host copy made
        on first access."""
        if self._S_host is None and self._S_dev is not None:
            self._S_host = self._S_dev.get().reshape(self.cri.shpS)
        return self._S_host

    @S.setter
    def S(self, value):
        self._S_host = value

    def getmin(self):
        """The minimiser (``ReturnX``: X, else Y) -- as a device view when ``resident``."""
        var = _lib.VAR_X if self.opt['ReturnX'] else _lib.VAR_Y
        if getattr(self, '_resident', False):
            from ..device import DeviceArray
            shp = self.cri.shpX[1:] if getattr(self, '_dim1', False) else self.cri.shpX
            return DeviceArray(shp, self.dtype, ptr=self._dev.device_ptr(var), base=self)
        return self._fetch(var)

    # -- device plumbing --------------------------------------------------------
    def _new_handle(self):
        H, W = self.cri.Nv
        self._dev = _lib.Solver(H, W, self.cri.C, self.cri.K, self.cri.M, self.dtype,
                                device=self._device, stream=self._stream, Cd=self.cri.Cd,
                                depth=self._dim3[0] if getattr(self, '_dim3', None) else 1)
        self._cache = {}
        self._no_x = getattr(self, '_no_x', False)   # promise that X / Xf will not be read
        self._u_scale = 1.0      # pending `U /= rsf` (admm.py:573), applied lazily
        self._sums = [0.0] * _lib.OUT_COUNT
        self._wl1_scalar = 1.0
        self._wl21_scalar = 1.0

    def _touch(self, *variables):
        """Forget cached host copies of device arrays that were just rewritten."""
        for v in variables:
            self._cache.pop(v, None)

    def _fetch(self, var):
        if var not in self._cache:
            a = self._dev.download(var)
            if var == _lib.VAR_U and self._u_scale != 1.0:
                a *= a.dtype.type(self._u_scale)
            if getattr(self, '_dim1', False) and a.ndim >= 2 and a.shape[0] == 1:
                a = a[0]       # (dimN = 1: the unit axis stays inside)
            if getattr(self, '_dim3', None):
                a = self._unfold(a)
            self._cache[var] = a
        return self._cache[var]

    def _store(self, var, value):
        if value is None:      # ADMM.__init__ convention `self.X = None`
            return
        value = np.asarray(value)
        if getattr(self, '_dim1', False) and value.ndim == len(self.cri.shpX) - 1:
            value = value[np.newaxis]
        if getattr(self, '_dim3', None) and value.ndim == len(self.cri.shpX) + 1:
            value = self._fold(value)
        self._dev.upload(var, value)
        if var == _lib.VAR_U:
            self._u_scale = 1.0
        self._touch(var)

    def init_state(self, yshape, ushape):
        """Y0 / U0 handling of sporco/admm/admm.py:262-272; device arrays start
        at zero.  (A Y0 without U0 gets U0 = (lmbda/rho) sign(Y0) once lmbda is
        known -- the intent documented at cbpdn.py:601-610.)"""
        if self.opt['Y0'] is not None:
            self.Y = np.asarray(self.opt['Y0']).astype(self.dtype, copy=True)
        if self.opt['U0'] is not None:
            self.U = np.asarray(self.opt['U0']).astype(self.dtype, copy=True)

    def __getstate__(self):
        state = self.__dict__.copy()
        for key in ('_dev', '_cache'):
            state.pop(key, None)
        state['_S_host'] = self.S          # (a device-resident signal travels as a host copy)
        state['_S_dev'] = None
        # raw device contents: U is saved WITHOUT the pending scale (kept in
        # _u_scale) so that a restored solver continues bit-identically
        state['_saved_arrays'] = {v: self._dev.download(v)
                                  for v in (_lib.VAR_Y, _lib.VAR_U, _lib.VAR_X)}
        state['_stream'] = None
        return state

    def __setstate__(self, state):
        saved = state.pop('_saved_arrays')
        u_scale = state.get('_u_scale', 1.0)
        if 'S' in state:       # pickled before `S` became a property: the key would be shadowed
            state.setdefault('_S_host', state.pop('S'))
        state.setdefault('_S_dev', None)
        self.__dict__.update(state)
        self._new_handle()
        self._dev.set_signal(self.S)
        self.setdict()
        self._upload_weights()
        for v, a in saved.items():
            self._store(v, a)
        self._u_scale = u_scale

    # -- dictionary ---------------------------------------------------------------
    def setdict(self, D=None):
        """Set the dictionary (internal layout, support (dH, dW) <= (H, W)):
        Df and the Sherman-Morrison denominators are rebuilt on device."""
        if D is not None:
            D = np.asarray(D, dtype=self.dtype)
            if getattr(self, '_dim1', False) and D.ndim == len(self.cri.shpD) - 1:
                D = D[np.newaxis]
            if getattr(self, '_dim3', None) and D.shape[0] != self.cri.Nv[0]:
                # (a dictionary in the reference's dimN = 3 shapes: zero-padded to the volume, folded)
                Dz, Hs = self._dim3
                D = D.reshape(D.shape[0:3] + (1, 1, D.shape[-1]))
                D = self._fold(cr.zpad(D, (Dz, Hs, self.cri.Nv[1])))
            self.D = D
        self._dev.set_dict(self.D)
        self._touch(_lib.VAR_DF)
        self.c = None

    def getcoef(self):
        return self.getmin()

    # -- parameters handed to the device --------------------------------------------
    def _flags(self):
        f = 0
        if self.opt['NonNegCoef']:
            f |= _lib.FLAG_NONNEG
        if self.opt['NoBndryCross']:
            f |= _lib.FLAG_NOBNDRY
        if self._needs_residuals():
            f |= _lib.FLAG_RESID
        if not self.opt['FastSolve']:
            f |= _lib.FLAG_OBJ
            if self.opt['gEvalY']:
                f |= _lib.FLAG_GEVAL_Y
            if not self.opt['fEvalX']:
                f |= _lib.FLAG_FEVAL_Y
        if self.opt['LinSolveCheck']:
            f |= _lib.FLAG_XRRS
        if self._no_x:
            f |= _lib.FLAG_NO_X
        if self._ams_mask is not None:
            f |= _lib.FLAG_AMS
        return f

    def _lmbda_eff(self):
        return 0.0

    def _mu_eff(self):
        return 0.0

    def _params(self, extra_flags=0):
        p = _lib.AdmmParams()
        p.rho = float(self.rho)
        p.lmbda = float(self._lmbda_eff())
        p.mu = float(self._mu_eff())
        p.rlx = float(self.rlx)
        p.u_scale = float(self._u_scale)
        p.flags = self._flags() | extra_flags
        p.dH, p.dW = int(self.D.shape[0]), int(self.D.shape[1])
        return p

    def _upload_weights(self):
        self._upload_ams()

    _ams_mask = None

    def _set_ams(self, W):
        """Switch on the additive-mask-simulation treatment of the last filter
        (the impulse appended by :class:`AddMaskSim`) with mask ``W`` in its internal
        5-D shape (cnvrep.mskWshape)."""
        self._ams_mask = np.asarray(W)
        self._upload_ams()

    def _upload_ams(self):
        if self._ams_mask is None:
            return
        H, W_, C, N, _ = self.cri.shpX
        # the reference zeroes `Yi[np.where(W.astype(bool))]` (cbpdn.py:2393): fancy
        # indexing with W's own index tuples, so an axis that W merely broadcasts over
        # addresses index 0 only; expanding the mask the same way keeps that behaviour.
        # (Multi-channel dictionary: one impulse slice per channel, the mask's channels
        # already moved to the last axis by AddMaskSim.)
        full = np.zeros((H, W_, C, N, self.cri.Cd), dtype=self.dtype)
        full[np.where(self._ams_mask.astype(bool))] = 1.0
        self._dev.set_ams_mask(full)

    # -- iteration: fused when nothing is overridden -----------------------------------
    def _fused_ok(self):
        base = type(self)._fused_base
        if base is None:
            return False
        for name in self._hook_names:
            if name in self.__dict__:
                return False
            if getattr(type(self), name) is not getattr(base, name):
                return False
        return True

    def _check_device_only_hooks(self):
        admm.refuse_step_overrides(self, self._device_only_names)

    def iteration(self):
        if not self._fused_ok():
            return super(GenericConvBPDN, self).iteration()
        p = self._params()
        if self._reducer is None:
            self._sums = self._dev.admm_iter(p)
        else:
            self._sums = self._reducer.admm_iter(self._dev, p)
        self._u_scale = 1.0
        self._touch(_lib.VAR_X, _lib.VAR_Y, _lib.VAR_U, _lib.VAR_XF)
        self._set_xrrs()
        if not self._needs_residuals():
            return None
        self.timer.stop('solve_wo_rsdl')
        res = self.compute_residuals()
        self.timer.start('solve_wo_rsdl')
        return res

    def finish_solve(self):
        self._dev.sync()

    # -- the whole loop on the device, when nothing on the host needs to see the iterations --
    def _device_loop_ok(self):
        """The device-driven solve (``sporco_amd_csc_admm_run``) replaces the loop of
        :meth:`ADMM.solve` when no callback, status display or overridden step has to run on
        the host between iterations; anything else keeps the per-iteration loop."""
        o = self.opt
        cls = type(self)
        # host methods the device loop never calls: an override of any of them (class or
        # instance) has to keep the per-iteration loop, or it would silently not run
        # (rhochange is the reference's documented hook, admm.py:575)
        host_only = (('solve', GenericConvBPDN), ('iteration', GenericConvBPDN),
                     ('finish_solve', GenericConvBPDN), ('rhochange', GenericConvBPDN),
                     ('update_rho', admm.ADMM), ('rho_scale_factor', admm.ADMM),
                     ('display_start', admm.ADMM), ('display_status', admm.ADMM),
                     ('display_end', admm.ADMM), ('iteration_stats', admm.ADMM))
        for name, base in host_only:
            if name in self.__dict__:
                return False
            if hasattr(base, name) and getattr(cls, name) is not getattr(base, name):
                return False
        return (self._fused_ok() and o['Callback'] is None and not o['Verbose'] and
                o['IterTimer'] == 'solve' and o['MaxMainIter'] > 0)

    def solve(self):
        """:meth:`ADMM.solve` (sporco/admm/admm.py:293-389).  When the iterations need no host
        involvement they run as one device-driven call: the residuals, the rho schedule and the
        stopping test are evaluated on the device after every iteration and the statistics are
        assembled from the per-iteration records afterwards (identical values; ``Time`` from
        the device clock; the ``solve_wo_func`` / ``solve_wo_rsdl`` timers then equal
        ``solve``)."""
        self._check_device_only_hooks()
        if not self._device_loop_ok():
            return super(GenericConvBPDN, self).solve()
        o = self.opt
        ctrl = _lib.AdmmCtrl()
        ctrl.abs_tol, ctrl.rel_tol = float(o['AbsStopTol']), float(o['RelStopTol'])
        ctrl.sqrt_nc, ctrl.sqrt_nx = float(np.sqrt(self.Nc)), float(np.sqrt(self.Nx))
        ctrl.rho_tau, ctrl.rho_mu, ctrl.rho_xi = \
            float(self.rho_tau), float(self.rho_mu), float(self.rho_xi)
        ctrl.auto_rho = int(bool(o['AutoRho', 'Enabled']))
        ctrl.period = int(o['AutoRho', 'Period'])
        ctrl.auto_scaling = int(bool(o['AutoRho', 'AutoScaling']))
        ctrl.std_residuals = int(bool(o['AutoRho', 'StdResiduals']))
        ctrl.need_residuals = int(self._needs_residuals())
        ctrl.k0, ctrl.max_iter, ctrl.lookahead = int(self.k), int(o['MaxMainIter']), 0
        all_timers = ['solve', 'solve_wo_func', 'solve_wo_rsdl']
        t_before = self.timer.elapsed('solve')
        self.timer.start(all_timers)
        reduce = None
        if self._reducer is not None:
            # (a callable: the per-iteration hook of TorchReducer; False: NativeReducer attached its
            # communicator and the library reduces by itself; None: no device-side reduction)
            reduce = self._reducer.device_sum_hook(self._dev)
        res = None
        if self._reducer is None or reduce is not None:
            res = self._dev.admm_run(self._params(), ctrl, reduce or None)
        if res is None:
            self.timer.stop(all_timers)
            return super(GenericConvBPDN, self).solve()
        recs, rho, u_scale = res
        self.timer.stop(all_timers)
        rdt = real_dtype(self.dtype).type
        fast = bool(o['FastSolve'])
        for rec in recs:
            self.k = int(rec.k)
            if not fast:
                self._sums = list(rec.sums)
                self.rho = rdt(rec.rho)
                self._set_xrrs()
                tk = t_before + rec.seconds
                tpl = (self.k,) + self.eval_objfn() + (rec.r, rec.s, rec.epri, rec.edua, self.rho) \
                    + self.itstat_extra() + (tk,)
                self.itstat.append(type(self).IterationStats(*tpl))
        if recs:
            self._sums = list(recs[-1].sums)
        self.rho = rdt(rho)
        self._u_scale = float(u_scale)
        self._touch(_lib.VAR_X, _lib.VAR_Y, _lib.VAR_U, _lib.VAR_XF)
        self.k += 1
        return self.getmin() if getattr(self, '_return_min', True) else None

    def _set_xrrs(self):
        if self.opt['LinSolveCheck']:
            s = self._sums
            nrm = max(np.sqrt(s[_lib.OUT_XRRS_AX2]), np.sqrt(s[_lib.OUT_XRRS_B2]))
            self.xrrs = 0.0 if nrm == 0.0 else np.sqrt(s[_lib.OUT_XRRS_D2]) / nrm
        else:
            self.xrrs = None

    # -- staged steps (one device call each; same roles as the reference methods) ---------
    def save_yprev(self):
        self._dev.copy(_lib.VAR_YPREV, _lib.VAR_Y)
        self._touch(_lib.VAR_YPREV)

    def xstep(self):
        """Y - U -> rfftn -> Sherman-Morrison -> irfftn (cbpdn.py:267-293)."""
        out = self._dev.admm_xstep(self._params())
        if self._reducer is not None:       # image shards: the sums of the other ranks
            out = self._reducer.sum(out)
        for slot in (_lib.OUT_DFID, _lib.OUT_XRRS_D2, _lib.OUT_XRRS_AX2, _lib.OUT_XRRS_B2,
                     _lib.OUT_RGR):
            self._sums[slot] = out[slot]
        self._touch(_lib.VAR_X, _lib.VAR_XF)
        self._set_xrrs()

    def relax_AX(self):
        """AX = alpha X + (1 - alpha) Y; AXnr is X itself (admm.py:877-885)."""
        self._dev.admm_relax(self.rlx)
        self._touch(_lib.VAR_AX)

    @property
    def AXnr(self):
        return self.X

    def ystep(self):
        """NonNegCoef / NoBndryCross enforcement only (cbpdn.py:297-311): the
        regulariser-specific shrinkage is added by derived classes."""
        self._ystep_device(0.0, 0.0, 0)

    def _ystep_device(self, lmbda, mu, extra_flags):
        p = self._params(extra_flags)
        p.lmbda, p.mu = float(lmbda), float(mu)
        self._dev.admm_ystep(p)
        self._touch(_lib.VAR_Y)

    def ustep(self):
        """U += AX - Y (admm.py:434-437)."""
        self._dev.admm_ustep(self._params())
        self._u_scale = 1.0
        self._touch(_lib.VAR_U)

    def residual_norms(self):
        if not self._fused_ok():
            out = self._dev.admm_stats(self._params(self._stats_flags()))
            if self._reducer is not None:
                out = self._reducer.sum(out)
            for slot in (_lib.OUT_R2, _lib.OUT_S2, _lib.OUT_AX2, _lib.OUT_Y2, _lib.OUT_U2,
                         _lib.OUT_L1, _lib.OUT_L21):
                self._sums[slot] = out[slot]
            if not self.opt['fEvalX']:
                self._sums[_lib.OUT_DFID] = out[_lib.OUT_DFID]
                self._sums[_lib.OUT_RGR] = out[_lib.OUT_RGR]
        s = self._sums
        rho = float(self.rho)
        nr = np.sqrt(s[_lib.OUT_R2])
        ns = rho * np.sqrt(s[_lib.OUT_S2])
        rn = max(np.sqrt(s[_lib.OUT_AX2]), np.sqrt(s[_lib.OUT_Y2]))
        sn = rho * np.sqrt(s[_lib.OUT_U2])
        return nr, ns, rn, sn

    def _stats_flags(self):
        return 0

    def rescale_u(self, rsf):
        """Defer ``U /= rsf`` (admm.py:573): the factor rides along as
        ``u_scale`` and is applied by the next kernels that read U."""
        self._u_scale = self._u_scale / float(rsf)
        self._touch(_lib.VAR_U)

    # -- objective --------------------------------------------------------------------------
    def eval_objfn(self):
        dfd = self.obfn_dfd()
        reg = self.obfn_reg()
        return (dfd + reg[0], dfd) + reg[1:]

    def obfn_dfd(self):
        """(1/2)||sum_m Df Xf - Sf||^2 by half-spectrum Parseval (cbpdn.py:337-344);
        the sum comes out of the Sherman-Morrison kernel as a by-product."""
        return self._sums[_lib.OUT_DFID] / 2.0

    def obfn_reg(self):
        raise NotImplementedError()

    def itstat_extra(self):
        return (self.xrrs,)

    def rhochange(self):
        pass

    def reconstruct(self, X=None, device=False):
        """irfftn(sum_m Df * rfftn(X)), X defaulting to Y (cbpdn.py:373-380).  ``device=True``:
        the result stays in HBM (a :class:`sporco_amd.device.DeviceArray` (H, W, C, N))."""
        if device:
            from ..device import DeviceArray
            if X is not None:
                raise NotImplementedError("reconstruct(device=True) reconstructs from Y")
            H, W_ = self.cri.Nv
            out = DeviceArray((H, W_, self._dev.Cs, self.cri.K), self.dtype)
            self._dev.reconstruct_dev(_lib.VAR_Y, out.ptr)
            return out
        if X is None:
            var = _lib.VAR_Y
        else:
            X = np.asarray(X, dtype=self.dtype)
            if getattr(self, '_dim1', False) and X.ndim == len(self.cri.shpX) - 1:
                X = X[np.newaxis]
            if getattr(self, '_dim3', None) and X.ndim == len(self.cri.shpX) + 1:
                X = self._fold(X)
            self._dev.upload(_lib.VAR_AX, X)
            self._touch(_lib.VAR_AX)
            var = _lib.VAR_AX
        r = self._dev.reconstruct(var)[..., 0]
        if getattr(self, '_dim3', None):
            return self._unfold(r)
        return r[0] if getattr(self, '_dim1', False) else r

    # -- per-kernel timing ---------------------------------------------------------------------
    def profile(self, enable=True):
        self._dev.profile(enable)

    def profile_read(self):
        return self._dev.profile_read()


class ConvBPDN(GenericConvBPDN):
    r"""Convolutional BPDN: minimise (1/2)||sum_m d_m * x_m - s||_2^2 +
    lambda sum_m ||x_m||_1 by ADMM (reference class: sporco/admm/cbpdn.py:386-630).

    IterationStats fields: ``Iter, ObjFun, DFid, RegL1, PrimalRsdl, DualRsdl,
    EpsPrimal, EpsDual, Rho, XSlvRelRes, Time``.
    """

    _dim1_ok = True


    class Options(GenericConvBPDN.Options):
        """Adds ``L1Weight`` (cbpdn.py:494-495)."""

        defaults = copy.deepcopy(GenericConvBPDN.Options.defaults)
        defaults.update({'L1Weight': 1.0})

        def __init__(self, opt=None):
            GenericConvBPDN.Options.__init__(self, {} if opt is None else opt)

    itstat_fields_objfn = ('ObjFun', 'DFid', 'RegL1')
    hdrtxt_objfn = ('Fnc', 'DFid', u'Regℓ1')
    hdrval_objfun = {'Fnc': 'ObjFun', 'DFid': 'DFid', u'Regℓ1': 'RegL1'}

    def __new__(cls, D=None, S=None, *args, **kwargs):
        # complex-valued signal / dictionary (cbpdn.py:209-217): the solver of cbpdn_cplx.py, which
        # runs the real machinery on the (re, im) channel pair
        if cls is ConvBPDN and isinstance(D, np.ndarray) and isinstance(S, np.ndarray) and \
                (np.iscomplexobj(D) or np.iscomplexobj(S)):
            from .cbpdn_cplx import ComplexConvBPDN
            return ComplexConvBPDN(D, S, *args, **kwargs)
        return super(ConvBPDN, cls).__new__(cls)

    def __init__(self, D, S, lmbda=None, opt=None, dimK=None, dimN=2, **backend):
        if opt is None:
            opt = ConvBPDN.Options()
        self.set_dtype(opt, S.dtype)
        super(ConvBPDN, self).__init__(D, S, opt, dimK, dimN, **backend)
        opt = self.opt     # (dimN = 1 / 3: a private copy with the array-valued entries reshaped)
        rdt = real_dtype(self.dtype).type
        if lmbda is None:
            # 0.1 * max |D^H s|  (cbpdn.py:573-578), evaluated on device
            lmbda = 0.1 * self._dev.dhs_absmax()
            if self._reducer is not None:
                lmbda = self._reducer.max(lmbda)
        self.lmbda = rdt(lmbda)
        self.set_attr('rho', opt['rho'], dval=(50.0 * self.lmbda + 1.0), dtype=rdt,
                      reset=True)
        if self.lmbda != 0.0:
            rho_xi = float(1.0 + (18.3) ** (np.log10(self.lmbda) + 1.0))
        else:
            rho_xi = 1.0
        self.set_attr('rho_xi', opt['AutoRho', 'RsdlTarget'], dval=rho_xi, dtype=rdt,
                      reset=True)
        self.wl1 = np.asarray(opt['L1Weight'], dtype=real_dtype(self.dtype))
        self.wl1 = self.wl1.reshape(cr.l1Wshape(self.wl1, self.cri))
        self._upload_weights()
        if opt['Y0'] is not None and opt['U0'] is None:
            self.U = (self.lmbda / self.rho) * np.sign(self.Y)

    def _upload_weights(self):
        super(ConvBPDN, self)._upload_weights()
        if self.wl1.size == 1:
            self._wl1_scalar = float(self.wl1.ravel()[0])
            self._dev.set_l1_weight(None)
        else:
            self._wl1_scalar = 1.0
            self._dev.set_l1_weight(_broadcastable(self.wl1, self.cri.shpX))

    def _lmbda_eff(self):
        # a scalar L1Weight folds into the threshold: (lmbda/rho) * w
        return float(self.lmbda) * self._wl1_scalar

    def uinit(self, ushape):
        if self.opt['Y0'] is None:
            return np.zeros(ushape, dtype=self.dtype)
        return (self.lmbda / self.rho) * np.sign(self.Y)

    def ystep(self):
        """Y = prox_l1(AX + U, (lmbda/rho) wl1), then NonNeg / NoBndryCross
        (cbpdn.py:614-620)."""
        self._ystep_device(self._lmbda_eff(), 0.0, 0)

    def obfn_reg(self):
        rl1 = abs(self._wl1_scalar) * self._sums[_lib.OUT_L1]
        return (self.lmbda * rl1, rl1)


class ConvBPDNJoint(ConvBPDN):
    r"""ConvBPDN with an additional l2,1 term over the channel axis,
    mu ||{x_c,m}||_{2,1} (reference class: sporco/admm/cbpdn.py:636-807).

    IterationStats fields: ``Iter, ObjFun, DFid, RegL1, RegL21, PrimalRsdl,
    DualRsdl, EpsPrimal, EpsDual, Rho, XSlvRelRes, Time``.
    """

    # (a multi-channel dictionary leaves the coefficient maps without a channel axis: the
    # l2,1 term then groups single elements, as it does in the reference)
    _multichannel_dict_ok = True

    class Options(ConvBPDN.Options):
        """Adds ``L21Weight`` (cbpdn.py:719-720)."""

        defaults = copy.deepcopy(ConvBPDN.Options.defaults)
        defaults.update({'L21Weight': 1.0})

        def __init__(self, opt=None):
            ConvBPDN.Options.__init__(self, {} if opt is None else opt)

    itstat_fields_objfn = ('ObjFun', 'DFid', 'RegL1', 'RegL21')
    hdrtxt_objfn = ('Fnc', 'DFid', u'Regℓ1', u'Regℓ2,1')
    hdrval_objfun = {'Fnc': 'ObjFun', 'DFid': 'DFid', u'Regℓ1': 'RegL1',
                     u'Regℓ2,1': 'RegL21'}

    def __init__(self, D, S, lmbda=None, mu=0.0, opt=None, dimK=None, dimN=2, **backend):
        if opt is None:
            opt = ConvBPDN.Options()
        self.mu = None
        self.wl21 = np.asarray(1.0)      # (until the options have been through the dimN set-up)
        super(ConvBPDNJoint, self).__init__(D, S, lmbda, opt, dimK=dimK, dimN=dimN,
                                            **backend)
        self.mu = self.dtype.type(mu)
        self.wl21 = np.asarray(self.opt['L21Weight'] if 'L21Weight' in self.opt else 1.0,
                               dtype=self.dtype)
        self._upload_weights()

    def _upload_weights(self):
        super(ConvBPDNJoint, self)._upload_weights()
        w = np.asarray(self.wl21)
        if w.size == 1:
            self._wl21_scalar = float(w.ravel()[0])
            self._dev.set_l21_weight(None)
        else:
            self._wl21_scalar = 1.0
            H, W_, C, N, K = self.cri.shpX
            if w.ndim == 5:
                w5 = w
            elif w.ndim <= 4:
                # the reference needs wl21 to broadcast against (H, W, N, K):
                # align trailing axes, then insert the (singleton) channel axis
                w4 = w.reshape((1,) * (4 - w.ndim) + w.shape)
                w5 = w4[:, :, np.newaxis, :, :]
            else:
                raise ValueError("L21Weight has too many dimensions")
            self._dev.set_l21_weight(_broadcastable(w5, (H, W_, 1, N, K)))

    def _mu_eff(self):
        return 0.0 if self.mu is None else float(self.mu) * self._wl21_scalar

    def _flags(self):
        return super(ConvBPDNJoint, self)._flags() | _lib.FLAG_JOINT

    def _stats_flags(self):
        return _lib.FLAG_JOINT

    def ystep(self):
        """Y = prox_sl1l2(AX + U, (lmbda/rho) wl1, (mu/rho) wl21, axis=C)
        (cbpdn.py:785-794)."""
        self._ystep_device(self._lmbda_eff(), self._mu_eff(), _lib.FLAG_JOINT)

    def obfn_reg(self):
        rl1 = abs(self._wl1_scalar) * self._sums[_lib.OUT_L1]
        rl21 = self._wl21_scalar * self._sums[_lib.OUT_L21]
        return (self.lmbda * rl1 + self.mu * rl21, rl1, rl21)


class ConvBPDNGradReg(ConvBPDN):
    r"""ConvBPDN with an l2 penalty on the gradient of the coefficient maps,
    (mu/2) sum_i sum_m w_m ||G_i x_m||_2^2 (reference class:
    sporco/admm/cbpdn.py:992-1214).  The gradient term enters the X step only:
    the per-frequency system becomes (diag(mu w_m GHGf + rho) + a a^H) x = b and is
    solved by the diagonal Sherman-Morrison form of ``linalg.solvedbd_sm``
    (sporco/linalg.py:300-366) inside the same kernel.

    With a multi-channel dictionary the system has one rank-one term per channel and is
    solved by iterated Sherman-Morrison with the same diagonal (``linalg.solvemdbi_ism``
    with an array ``rho``, cbpdn.py:1181-1184).

    IterationStats fields: ``Iter, ObjFun, DFid, RegL1, RegGrad, PrimalRsdl,
    DualRsdl, EpsPrimal, EpsDual, Rho, XSlvRelRes, Time``.
    """

    _dim1_ok = False     # (its gradient term has a spatial structure: dimN = 2 only)


    _multichannel_dict_ok = True

    class Options(ConvBPDN.Options):
        """Adds ``GradWeight``: scalar, or one weight per filter (cbpdn.py:1059-1073)."""

        defaults = copy.deepcopy(ConvBPDN.Options.defaults)
        defaults.update({'GradWeight': 1.0})

        def __init__(self, opt=None):
            ConvBPDN.Options.__init__(self, {} if opt is None else opt)

    itstat_fields_objfn = ('ObjFun', 'DFid', 'RegL1', 'RegGrad')
    hdrtxt_objfn = ('Fnc', 'DFid', u'Regℓ1', u'Regℓ2∇')
    hdrval_objfun = {'Fnc': 'ObjFun', 'DFid': 'DFid', u'Regℓ1': 'RegL1',
                     u'Regℓ2∇': 'RegGrad'}

    def __init__(self, D, S, lmbda=None, mu=0.0, opt=None, dimK=None, dimN=2, **backend):
        if opt is None:
            opt = ConvBPDNGradReg.Options()
        self.set_dtype(opt, S.dtype)
        self.mu = self.dtype.type(mu)
        gw = opt['GradWeight']
        if hasattr(gw, 'ndim') and np.ndim(gw) > 0:
            # one weight per filter, broadcast along the filter axis (cbpdn.py:1134-1137)
            self.Wgrd = np.asarray(np.asarray(gw).reshape((1,) * (dimN + 2) + np.shape(gw)),
                                   dtype=self.dtype)
        else:
            self.Wgrd = np.asarray(gw, dtype=self.dtype)
        super(ConvBPDNGradReg, self).__init__(D, S, lmbda, opt, dimK=dimK, dimN=dimN,
                                              **backend)

    @property
    def GHGf(self):
        """Wgrd * sum_i |G_i|^2 on the half spectrum (cbpdn.py:1141-1143); host-side
        convenience, the kernels evaluate it from two separable tables."""
        H, W = self.cri.Nv
        gh = (2.0 - 2.0 * np.cos(2.0 * np.pi * np.arange(H) / H)) if H > 1 else np.ones(1)
        gw = (2.0 - 2.0 * np.cos(2.0 * np.pi * np.arange(W // 2 + 1) / W)) if W > 1 \
            else np.ones(1)
        g = (gh[:, np.newaxis] + gw[np.newaxis, :]).reshape(H, W // 2 + 1, 1, 1, 1)
        return self.Wgrd * g.astype(self.dtype)

    def _upload_weights(self):
        super(ConvBPDNGradReg, self)._upload_weights()
        if self.Wgrd.size == 1:
            self._wg_scalar = float(self.Wgrd.ravel()[0])
            self._dev.set_grad_weight(None)
        else:
            if self.Wgrd.size != self.cri.M:
                raise ValueError("GradWeight must be a scalar or hold one weight per filter")
            self._wg_scalar = 1.0
            self._dev.set_grad_weight(self.Wgrd.ravel())

    def _mu_eff(self):
        # a scalar GradWeight folds into mu: mu * (w GHGf)
        return float(self.mu) * self._wg_scalar

    def _flags(self):
        return super(ConvBPDNGradReg, self)._flags() | _lib.FLAG_GRADREG

    def obfn_reg(self):
        rl1 = abs(self._wl1_scalar) * self._sums[_lib.OUT_L1]
        rgr = self._wg_scalar * self._sums[_lib.OUT_RGR] / 2.0
        return (self.lmbda * rl1 + self.mu * rgr, rl1, rgr)


class ConvBPDNMaskDcpl(ConvBPDN):
    r"""ConvBPDN with a spatial mask in the data fidelity term, by mask decoupling: minimise
    (1/2)||W(sum_m d_m * x_m - s)||_2^2 + lambda sum_m ||x_m||_1 through the two-block
    constraint [D; I] x - [y0; y1] = [s; 0] (reference class: sporco/admm/cbpdn.py:2066-2283 on
    ConvTwoBlockCnstrnt :1401-1826 and admm.ADMMTwoBlockCnstrnt, sporco/admm/admm.py:989-1437).

    One iteration is one call of ``sporco_amd_csc_mdcpl_iter``: the X-step is the
    Sherman-Morrison solve of ConvBPDN with rho = 1 and the block-0 spectrum in the signal's
    place, block 1 (``y1``, ``u1``: coefficient sized) runs through the ConvBPDN epilogue
    kernel, block 0 (``y0``, ``u0``: signal sized) through its own small kernel; the host forms
    residuals, objective and the rho schedule from the returned sums.

    IterationStats fields: ``Iter, ObjFun, DFid, RegL1, PrimalRsdl, DualRsdl, EpsPrimal,
    EpsDual, Rho, XSlvRelRes, Time``.  Multi-channel dictionaries (``Cd > 1``): block 0 keeps the
    signal's channels and is swapped onto the filter axis for ``Y`` / ``U`` (cbpdn.py:1688-1722).
    """

    _dim1_ok = False


    _multichannel_dict_ok = True    # (X-step by linalg.solvemdbi_ism with rho = 1, cbpdn.py:1621-1626)
    _fused_base = None

    class Options(admm.ADMMEqual.Options):
        """ConvTwoBlockCnstrnt.Options (cbpdn.py:1428-1519) + ``L1Weight`` (:2123-2128): the
        ADMM base defaults (AutoRho off) with ``rho`` 1.0, ``RelaxParam`` 1.8, ``ReturnVar``
        'Y1'.  ``HighMemSolve`` is accepted and has no effect."""

        defaults = copy.deepcopy(admm.ADMMEqual.Options.defaults)
        defaults.update({'AuxVarObj': False, 'fEvalX': True, 'gEvalY': False,
                         'HighMemSolve': False, 'LinSolveCheck': False, 'NonNegCoef': False,
                         'NoBndryCross': False, 'RelaxParam': 1.8, 'rho': 1.0,
                         'ReturnVar': 'Y1', 'L1Weight': 1.0})

        def __init__(self, opt=None):
            admm.ADMMEqual.Options.__init__(self, {} if opt is None else opt)

    itstat_fields_objfn = ('ObjFun', 'DFid', 'RegL1')
    itstat_fields_extra = ('XSlvRelRes',)
    hdrtxt_objfn = ('Fnc', 'DFid', u'Regℓ1')
    hdrval_objfun = {'Fnc': 'ObjFun', 'DFid': 'DFid', u'Regℓ1': 'RegL1'}

    def __init__(self, D, S, lmbda, W=None, opt=None, dimK=None, dimN=2, **backend):
        if opt is None:
            opt = ConvBPDNMaskDcpl.Options()
        if opt['ReturnVar'] not in ('X', 'Y0', 'Y1'):
            raise ValueError(str(opt['ReturnVar']) + ' is not a valid value for option ReturnVar')
        # (the warm-start arrays are two-block arrays: kept away from ConvBPDN's own Y0 / U0
        # handling -- its U0 = (lmbda / rho) sign(Y0) rule is not this class's, whose dual
        # variable starts at zero, admm.py:286-289 -- and stored once the blocks exist)
        warm = {'Y0': opt['Y0'], 'U0': opt['U0']}
        if warm['Y0'] is not None or warm['U0'] is not None:
            opt = copy.deepcopy(opt)
            opt['Y0'] = None
            opt['U0'] = None
        super(ConvBPDNMaskDcpl, self).__init__(D, S, lmbda, opt, dimK=dimK, dimN=dimN, **backend)
        rdt = real_dtype(self.dtype).type
        # ADMM base values, not ConvBPDN's lambda-dependent ones (admm.py:245-253)
        self.set_attr('rho', opt['rho'], dval=1.0, dtype=rdt, reset=True)
        self.set_attr('rho_xi', opt['AutoRho', 'RsdlTarget'], dval=1.0, dtype=rdt, reset=True)
        # problem sizes of the two-block constraint (cbpdn.py:1568-1574)
        self.Nx = self.cri.M * self.cri.N * self.cri.K
        self.Nc = int(np.prod(self.cri.shpX)) + int(np.prod(self.cri.shpS))
        if self._reducer is not None:       # image shards: the global problem's sizes
            from ..dist import global_count
            self.Nx = global_count(self._reducer, self.Nx)
            self.Nc = global_count(self._reducer, self.Nc)
        if W is None:
            W = np.array([1.0], dtype=self.dtype)
        W = np.asarray(W)
        shp = (1,) * 5 if W.size == 1 else cr.mskWshape(W, self.cri)
        self.W = np.asarray(W.reshape(shp), dtype=self.dtype)
        self._upload_weights()
        self._dev.mdcpl_init(self.S)
        # warm start (admm.py:262-272): Y0 / U0 are the concatenated [block 0; block 1] arrays
        # of ADMMTwoBlockCnstrnt, (H, W, C, N, Cd + K) (cbpdn.py:1565-1574)
        for name, v0, v1 in (('Y0', _lib.VAR_MY0, _lib.VAR_Y), ('U0', _lib.VAR_MU0, _lib.VAR_U)):
            if warm[name] is not None:
                self._set_blocks(np.asarray(warm[name]), v0, v1)
                self.opt[name] = warm[name]
        s2 = float(np.linalg.norm(self.S)) ** 2
        if self._reducer is not None:
            s2 = self._reducer.sum([s2])[0]
        self._nrm_c = float(np.sqrt(s2))

    def _upload_weights(self):
        super(ConvBPDNMaskDcpl, self)._upload_weights()
        if hasattr(self, 'W'):
            H, Wd = self.cri.Nv
            self._dev.set_data_mask(_broadcastable(self.W, (H, Wd, self.cri.C, self.cri.K, 1)))

    @property
    def _y0swap(self):
        # block 0 has the shape of S, (.., C, K, 1); with a multi-channel dictionary its channel
        # axis is swapped onto the filter axis so that it concatenates with block 1 (.., 1, K, M)
        return self.cri.C > 1 and self.cri.Cd > 1

    def _set_blocks(self, A, var0, var1):
        """Store a concatenated two-block array: block 0 (the first Cd slices of the last axis,
        signal sized) and block 1 (coefficient sized)."""
        A = np.asarray(A, dtype=self.dtype)
        nb0 = self.cri.Cd
        shp = self.cri.shpX[:-1] + (nb0 + self.cri.M,)
        if A.size != int(np.prod(shp)):
            raise ValueError("array of shape %s is not a [block 0; block 1] array of shape %s"
                             % (A.shape, shp))
        A = A.reshape(shp)
        A0 = A[..., :nb0]
        if self._y0swap:
            A0 = np.swapaxes(A0, self.cri.axisC, self.cri.axisM)
        self._dev.upload(var0, np.ascontiguousarray(A0))
        self._dev.upload(var1, np.ascontiguousarray(A[..., nb0:]))
        self._touch(var0, var1)

    def __getstate__(self):
        state = self.__dict__.copy()
        for key in ('_dev', '_cache'):
            state.pop(key, None)
        state['_S_host'] = self.S
        state['_S_dev'] = None
        state['_stream'] = None
        # raw device contents of both blocks (U without the pending scale, kept in _u_scale)
        state['_saved_arrays'] = {v: self._dev.download(v)
                                  for v in (_lib.VAR_Y, _lib.VAR_U, _lib.VAR_X, _lib.VAR_MY0,
                                            _lib.VAR_MU0)}
        return state

    def __setstate__(self, state):
        saved = state.pop('_saved_arrays')
        u_scale = state.get('_u_scale', 1.0)
        self.__dict__.update(state)
        self._new_handle()
        self._dev.set_signal(self.S)
        self.setdict()
        self._upload_weights()
        self._dev.mdcpl_init(self.S)
        for v, a in saved.items():
            self._dev.upload(v, a)
        self._u_scale = u_scale

    # -- the two blocks -----------------------------------------------------------------------
    def var_y0(self):
        return self._dev.download(_lib.VAR_MY0)

    def var_y1(self):
        return self._fetch(_lib.VAR_Y)

    def block_sep0(self, Y):
        Y0 = Y[..., :self.cri.Cd]
        return np.swapaxes(Y0, self.cri.axisC, self.cri.axisM) if self._y0swap else Y0

    def block_sep1(self, Y):
        return Y[..., self.cri.Cd:]

    def block_cat(self, Y0, Y1):
        if self._y0swap:
            Y0 = np.swapaxes(Y0, self.cri.axisC, self.cri.axisM)
        return np.concatenate((Y0, Y1), axis=self.cri.axisM)

    @property
    def Y(self):
        """[y0; y1] concatenated on the filter axis, as the reference keeps it."""
        return self.block_cat(self.var_y0(), self.var_y1())

    @Y.setter
    def Y(self, value):
        if value is not None:
            raise NotImplementedError("the blocks of Y are device state; use var_y0 / var_y1")

    @property
    def U(self):
        u0 = self._dev.download(_lib.VAR_MU0)
        if self._u_scale != 1.0:
            u0 *= u0.dtype.type(self._u_scale)
        return self.block_cat(u0, self._fetch(_lib.VAR_U))

    @U.setter
    def U(self, value):
        if value is not None:
            raise NotImplementedError("the blocks of U are device state")

    def getmin(self):
        rv = self.opt['ReturnVar']
        return self.X if rv == 'X' else (self.var_y0() if rv == 'Y0' else self.var_y1())

    # -- iteration ----------------------------------------------------------------------------
    def iteration(self):
        admm.refuse_step_overrides(self)
        p = self._params()
        flags = 0
        if self.opt['NonNegCoef']:
            flags |= _lib.FLAG_NONNEG
        if self.opt['NoBndryCross']:
            flags |= _lib.FLAG_NOBNDRY
        if self._needs_residuals():
            flags |= _lib.FLAG_RESID
        if not self.opt['FastSolve']:
            flags |= _lib.FLAG_OBJ
        if self.opt['AuxVarObj']:
            flags |= _lib.FLAG_GEVAL_Y
        if self.opt['LinSolveCheck']:
            flags |= _lib.FLAG_XRRS
        p.flags = flags
        self._sums = self._dev.mdcpl_iter(p)
        if self._reducer is not None:
            self._sums = self._reducer.sum(self._sums)
        self._u_scale = 1.0
        self._touch(_lib.VAR_X, _lib.VAR_Y, _lib.VAR_U, _lib.VAR_XF)
        self._set_xrrs()
        if not self._needs_residuals():
            return None
        self.timer.stop('solve_wo_rsdl')
        res = self.compute_residuals()
        self.timer.start('solve_wo_rsdl')
        return res

    def residual_norms(self):
        """admm.py:1404-1437 with the dual residual of cbpdn.py:1814-1824."""
        s = self._sums
        rho = float(self.rho)
        nr = np.sqrt(s[_lib.OUT_R2] + s[_lib.OUT_L21])
        ns = rho * np.sqrt(s[_lib.OUT_S2])
        rn = max(np.sqrt(s[_lib.OUT_AX2] + s[_lib.OUT_RGR]),
                 np.sqrt(s[_lib.OUT_Y2] + s[_lib.OUT_CNSTR]), self._nrm_c)
        sn = rho * np.sqrt(s[_lib.OUT_U2] + s[_lib.OUT_CGIT])
        return nr, ns, rn, sn

    def eval_objfn(self):
        """(1/2)||W g0||^2 + lmbda ||wl1 g1||_1 (cbpdn.py:2251-2275)."""
        g0v = self._sums[_lib.OUT_DFID] / 2.0
        g1v = abs(self._wl1_scalar) * self._sums[_lib.OUT_L1]
        return (g0v + self.lmbda * g1v, g0v, g1v)

    def reconstruct(self, X=None):
        """irfftn(sum_m Df * rfftn(X)), X defaulting to the X variable (cbpdn.py:1801-1810)."""
        if X is None:
            return self._dev.reconstruct(_lib.VAR_X)[..., 0]
        return super(ConvBPDNMaskDcpl, self).reconstruct(X)


class AddMaskSim(object):
    """Boundary / missing-data masking by additive mask simulation: wrapper about a
    ConvBPDN-family object of this module with an impulse filter appended to the
    dictionary (reference class: sporco/admm/cbpdn.py:2287-2485, same constructor,
    methods and attributes; ``b.cbpdn`` is the inner solver).

    The reference installs Python replacements for the inner object's ``ystep`` and
    ``obfn_gvar``.  Here the inner solver is told about the mask instead
    (``SPORCO_AMD_FLAG_AMS``): its y step and regulariser sums treat the impulse slice
    on the device, so the iteration stays one fused call.
    """

    def __init__(self, cbpdnclass, D, S, W, *args, **kwargs):
        dimK = kwargs.get('dimK', None)
        dimN = kwargs.get('dimN', 2)
        if dimN != 2:
            raise NotImplementedError("AddMaskSim: the masked kernels know two spatial axes (dimN = 2)")
        self.cri = cr.CSC_ConvRepIndexing(D, S, dimK=dimK, dimN=dimN)
        if not hasattr(cbpdnclass, '_set_ams'):
            raise TypeError("AddMaskSim wraps the solver classes of sporco_amd.admm.cbpdn")
        # impulse filter -- one per channel of a multi-channel dictionary -- appended to the
        # dictionary (cbpdn.py:2337-2346)
        if self.cri.Cd == 1:
            self.imp = np.zeros(D.shape[0:dimN] + (1,))
            self.imp[(0,) * dimN] = 1.0
        else:
            self.imp = np.zeros(D.shape[0:dimN] + (self.cri.Cd,) * 2)
            for c in range(self.cri.Cd):
                self.imp[(0,) * dimN + (c, c)] = 1.0
        Di = np.concatenate((D, self.imp), axis=D.ndim - 1)
        self.cbpdn = cbpdnclass(Di, S, *args, **kwargs)
        self.IterationStats = self.cbpdn.IterationStats
        self.W = np.asarray(W.reshape(cr.mskWshape(W, self.cri)), dtype=self.cbpdn.dtype)
        # the mask's channels go where the per-channel impulse filters are (cbpdn.py:2358-2364)
        if self.cri.Cd > 1 and self.W.shape[self.cri.dimN] > 1:
            self.W = np.swapaxes(self.W, self.cri.axisC, self.cri.axisM)
        self.cbpdn._set_ams(self.W)

    def solve(self):
        """Solve with the inner object; the AMS component is stripped from the result."""
        Xi = self.cbpdn.solve()
        self.timer = self.cbpdn.timer
        self.itstat = self.cbpdn.itstat
        return Xi[self.index_primary()]

    def setdict(self, D=None):
        imp = self.imp.reshape(self.imp.shape[:-1] + (1,) * (D.ndim - self.imp.ndim) +
                               self.imp.shape[-1:])
        self.cbpdn.setdict(np.concatenate((D, imp), axis=D.ndim - 1))

    def getcoef(self):
        return self.cbpdn.getcoef()[self.index_primary()]

    def index_primary(self):
        return np.s_[..., 0:-self.cri.Cd]

    def index_addmsk(self):
        return np.s_[..., -self.cri.Cd:]

    def reconstruct(self, X=None):
        """Reconstruction from the primary component only (cbpdn.py:2457-2477)."""
        if X is None:
            X = self.cbpdn.Y[self.index_primary()]
        X = np.asarray(X).reshape(self.cri.shpX[:-1] + (-1,))
        # a zero impulse slice: the inner reconstruction then sums the primary filters only
        Xi = np.concatenate((X, np.zeros(X.shape[:-1] + (self.cri.Cd,), dtype=X.dtype)),
                            axis=-1)
        return self.cbpdn.reconstruct(Xi)

    def getitstat(self):
        return self.cbpdn.getitstat()


def _broadcastable(w, full_shape):
    """Return ``w`` as a 5-D array whose axes are each 1 or the full extent."""
    w = np.asarray(w)
    if w.ndim != 5:
        raise ValueError("weight array must be 5-dimensional after reshaping")
    tgt = tuple(full_shape)
    for ws, ts in zip(w.shape, tgt):
        if ws not in (1, ts):
            raise ValueError("weight array of shape %s cannot broadcast to %s" %
                             (w.shape, tgt))
    return np.ascontiguousarray(w)


GenericConvBPDN._fused_base = None
ConvBPDN._fused_base = ConvBPDN
ConvBPDNJoint._fused_base = ConvBPDNJoint
ConvBPDNGradReg._fused_base = ConvBPDNGradReg
