"""ADMM dictionary updates on the GPU: convolutional constrained MOD in its consensus form
and with a single dictionary copy (iterated Sherman-Morrison / conjugate gradient X-steps).

Drop-ins for ``sporco.admm.ccmod.ConvCnstrMOD_Consensus`` (sporco/admm/ccmod.py:605-908, on
``admm.ADMMConsensus``, sporco/admm/admm.py:1441-1707), ``ConvCnstrMOD_IterSM`` (:433-505) and
``ConvCnstrMOD_CG`` (:511-601, both on ``ConvCnstrMODBase`` :103-429): same constructors,
Options trees, IterationStats fields and methods (``setcoef / getdict / solve / getitstat``).

Each image keeps its own copy X_n of the (zero-padded) dictionary and a dual U_n, both
resident on the device; the consensus variable Y is the dictionary.  One iteration is one call
into the C ABI (``sporco_amd_csc_cns_iter``): the per-image X-step is the Sherman-Morrison
solve of the sparse coding path with the roles of dictionary and coefficients swapped
(ccmod.py:766-778), followed by the mean over images + constraint projection and the dual
update; the host forms residuals, tolerances and the rho schedule from the returned sums.
"""

import copy

import numpy as np

from . import admm
from .. import _lib
from .. import cnvrep as cr
from .cbpdn import _reshaped_options

__all__ = ['ConvCnstrMOD_Consensus', 'ConvCnstrMOD_IterSM', 'ConvCnstrMOD_CG', 'ConvCnstrMOD',
           'ConvCnstrMODOptions']


class _DeviceDStep(object):
    """What the ADMM dictionary updates of this module share whatever their splitting: the
    device handle (own or the sparse coding step's), coefficient maps, the dictionary Y =
    ``SPORCO_AMD_VAR_DX`` with its read-back, the two objective terms from the sums of the last
    device call, the deferred ``U /= rsf``."""

    itstat_fields_objfn = ('DFid', 'Cnstr')
    itstat_fields_extra = ('XSlvRelRes',)
    hdrtxt_objfn = ('DFid', 'Cnstr')
    hdrval_objfun = {'DFid': 'DFid', 'Cnstr': 'Cnstr'}

    # Complex-valued coefficient maps, signals and dictionary (the reference takes them through
    # its fftn path, ccmod.py:219-231; tests/admm/test_ccmod.py:49-140): the real and the imaginary
    # part are the two channels of a real problem with a two-channel dictionary whose coefficient
    # maps carry the channels -- constraint projection, residuals and dual update of the complex
    # problem are those of that real problem as they stand (a complex l2 norm is the norm over
    # the pair) -- and the handle pairs the channels' spectra in its linear solves
    # (``SPORCO_AMD_MODE_COMPLEX_PAIR``, include/sporco_amd.h).  ``dtype`` stays the real working
    # type; ``cdtype`` is the complex one of the arrays the caller sees.
    _cplx = False

    # dimN = 1 (signals): run as images with a unit first axis, which the arrays the caller sees
    # (Y, X, U, getdict(), reconstruct()) do not show -- as in admm/cbpdn.py and pgm/ccmod.py
    _dim1 = False

    def _signal_setup(self, S, dsz, opt, dimN):
        if dimN != 1:
            return S, dsz, opt, dimN
        from ..pgm.ccmod import _dsz_unit_axis
        if np.iscomplexobj(S):
            raise NotImplementedError("complex-valued dictionary update: dimN = 2")
        self._dim1 = True
        # (an array in the reference's dimN = 1 shape (N, C, K, M) has four axes, the internal one five)
        opt = _reshaped_options(opt, ('Y0', 'U0'), lambda a: a[np.newaxis], when=lambda a: a.ndim == 4)
        return np.asarray(S)[np.newaxis], _dsz_unit_axis(dsz), opt, 2

    # dimN = 3 (volumes; consensus update only): the first two axes folded, on a volume handle whose
    # projections crop in three axes (admm/cbpdn.py, pgm/ccmod.py)
    _dim3 = None

    def _volume_setup(self, S, dsz, opt, dimK, dimN, reducer):
        if dimN != 3:
            return S, dsz, opt, dimK, dimN
        if isinstance(dsz[0], (list, tuple)) or reducer is not None or np.iscomplexobj(S):
            raise NotImplementedError("dimN = 3: one filter support, real data, no image shards")
        S = np.asarray(S)
        c3 = cr.CDU_ConvRepIndexing(dsz, S, dimK=dimK, dimN=3)
        if c3.Cd > 1:
            raise NotImplementedError("dimN = 3: single-channel dictionary")
        if opt['ZeroMean']:
            raise NotImplementedError("dimN = 3 with ZeroMean (see pgm/ccmod.py)")
        self._dim3 = (int(c3.Nv[0]), int(c3.Nv[1]))
        self._dsz3, self._cri3 = tuple(int(v) for v in dsz), c3
        opt = _reshaped_options(opt, ('Y0', 'U0'), lambda a: cr.fold3(a, *self._dim3),
                                when=lambda a: a.ndim >= 6)
        S2 = cr.fold3(S.reshape(c3.shpS), *self._dim3)[..., 0]
        return S2, (self._dim3[0] * self._dim3[1], int(c3.Nv[2]), c3.M), opt, 1, 2

    def _drop1(self, a):
        if self._dim3:
            return cr.unfold3(a, *self._dim3)
        return a[0] if self._dim1 else a

    def _add1(self, a, ndim):
        """The unit axis back on an array handed in with the reference's dimN = 1 shape (dimN = 3:
        the first two axes folded)."""
        a = np.asarray(a)
        if self._dim3:
            return cr.fold3(a, *self._dim3) if a.ndim == ndim + 1 else a
        return a[np.newaxis] if self._dim1 and a.ndim == ndim - 1 else a

    def _complex_setup(self, Z, S, dsz, opt, dimK, dimN, dev, reducer=None):
        """The arguments of the equivalent real problem when any of ``Z`` / ``S`` is complex
        (or ``DataType`` asks for it); the complex problem's indexing in ``self.cri_c``."""
        dt = opt['DataType']
        if not (np.iscomplexobj(S) or (Z is not None and np.iscomplexobj(Z))
                or (dt is not None and np.dtype(dt).kind == 'c')):
            return Z, S, dsz, dimK
        if dev is not None or reducer is not None:
            raise NotImplementedError("complex-valued dictionary update: not with a shared device "
                                      "handle or image shards")
        if isinstance(dsz[0], (list, tuple)):
            raise NotImplementedError("complex-valued dictionary update: one filter support")
        S = np.asarray(S)
        cri = cr.CDU_ConvRepIndexing(dsz, S, dimK=dimK, dimN=dimN)
        if cri.C > 1 or cri.Cd > 1:
            raise NotImplementedError("complex-valued dictionary update: single-channel signals "
                                      "and dictionary")
        if opt['fEvalX'] and type(self)._objective_at_y_only:
            raise NotImplementedError("complex-valued consensus update: objective at Y (AuxVarObj)")
        cdt = np.result_type(S.dtype if dt is None else np.dtype(dt), np.complex64)
        self._cplx = True
        self.cri_c = cri
        self.cdtype = np.dtype(cdt)
        self.dtype = np.dtype(np.float32 if cdt == np.complex64 else np.float64)
        Sc = S.reshape(cri.Nv + (1, cri.K))
        S2 = np.concatenate([Sc.real, Sc.imag], axis=2).astype(self.dtype)     # (H, W, 2, K)
        return Z, S2, tuple(dsz[0:dimN]) + (2, cri.M), 1

    def _to_pair(self, a):
        """Complex array with a unit channel axis (axis 2) -> real array with (re, im) channels."""
        if a is None or not self._cplx:
            return a
        a = np.asarray(a)
        if a.ndim < 3 or a.shape[2] != 1:
            raise ValueError("complex-valued array with a unit channel axis expected, got shape %s"
                             % (a.shape,))
        return np.concatenate([a.real, a.imag], axis=2).astype(self.dtype)

    def _from_pair(self, a):
        if not self._cplx:
            return a
        return (a[:, :, 0:1] + 1j * a[:, :, 1:2]).astype(self.cdtype)

    def _attach_device(self, S, dev, device, stream):
        """``self.S`` in the internal layout (one block per (channel, image): channels fold
        into the image axis for a single-channel dictionary, ccmod.py:695-699, :702-706) and
        ``self.dev``."""
        H, W = self.cri.Nv
        self._shared = dev is not None
        if self.cri.Cd > 1:
            # multi-channel dictionary (consensus update only): S keeps its channels, the
            # coefficient maps have none, one consensus block per image (ccmod.py:696-698)
            self.Nb = self.cri.K
            self.S = np.asarray(S.reshape(self.cri.shpS), dtype=self.dtype)
            if dev is None:
                self.dev = _lib.Solver(H, W, self.cri.C, self.cri.K, self.cri.M, self.dtype,
                                       device=device, stream=stream, Cd=self.cri.Cd)
                if self._cplx:
                    self.dev.set_hint(_lib.MODE_COMPLEX_PAIR, 1)
                self.dev.set_signal(self.S)
            else:
                if (dev.dims[:2] + (dev.Cs,) + dev.dims[3:]) != (H, W, self.cri.C, self.cri.K,
                                                                  self.cri.M) \
                        or dev.Cd != self.cri.Cd or dev.dtype != self.dtype:
                    raise ValueError("shared device solver has different dimensions")
                self.dev = dev
            self._cache = {}
            self._u_scale = 1.0
            self._sums = [0.0] * _lib.OUT_COUNT
            self.dev.set_filter_sizes(self.cri.fsz)
            return
        self.Nb = self.cri.C * self.cri.K
        self.S = np.asarray(S.reshape(self.cri.Nv + (1, self.Nb, 1)), dtype=self.dtype)
        vol = self._dim3[0] if self._dim3 else 1
        if dev is None:
            self.dev = _lib.Solver(H, W, self.cri.C, self.cri.K, self.cri.M, self.dtype,
                                   device=device, stream=stream, depth=vol)
            self.dev.set_signal(self.S)
        else:
            if dev.dims != (H, W, self.cri.C, self.cri.K, self.cri.M) or dev.dtype != self.dtype \
                    or getattr(dev, 'depth', 1) != vol:
                raise ValueError("shared device solver has different dimensions")
            self.dev = dev
        self._cache = {}
        self._u_scale = 1.0
        self._sums = [0.0] * _lib.OUT_COUNT
        if self._dim3:
            self.dev.set_hint(_lib.VOLUME_FILTER_DEPTH, self._dsz3[0])
        else:
            # (multi-scale dsz: every filter's own support for the projections of this handle)
            self.dev.set_filter_sizes(self.cri.fsz)

    @property
    def Y(self):
        if _lib.VAR_DX not in self._cache:
            self._cache[_lib.VAR_DX] = self._drop1(self._from_pair(self.dev.download(_lib.VAR_DX)))
        return self._cache[_lib.VAR_DX]

    @Y.setter
    def Y(self, value):
        if value is None:
            return
        self.dev.upload(_lib.VAR_DX, np.asarray(self._to_pair(self._add1(value, 5)), dtype=self.dtype))
        self.dev.fft_var(_lib.VAR_DX, _lib.VAR_DXF)
        self._cache.pop(_lib.VAR_DX, None)

    def getmin(self):
        return self.Y

    def setcoef(self, Z):
        """Set the coefficient maps: Zf = rfftn(Z) (ccmod.py:311-327, :746-755)."""
        # (multi-channel dictionary: Nb = K and the maps have no channel axis, same reshape)
        Z = np.asarray(Z)
        if self._dim3 and Z.shape[0:2] == self._dim3:
            Z = cr.fold3(Z.reshape(self._cri3.shpX), *self._dim3)
        if self._cplx:
            # (the complex maps for reconstruct(); the device gets their (re, im) channel pair)
            self._Zc = Z.reshape(self.cri.Nv + (1, self.Nb, self.cri.M)).astype(self.cdtype)
            Z = np.concatenate([self._Zc.real, self._Zc.imag], axis=2)
        self._z_chan = self.cri.Cd > 1 and Z.size == self.cri.N * self.cri.Cd * self.Nb * self.cri.M
        if self._z_chan:
            # ... unless they carry the dictionary's channels: the reference's broadcasting then
            # solves Cd independent single-channel updates (tests/admm/test_ccmod.py:278-295).
            # The device takes them in its consensus blocks' layout (H, W, K, Cd, M).
            self.Z = np.asarray(Z.reshape(self.cri.Nv + (self.cri.Cd, self.Nb, self.cri.M)),
                                dtype=self.dtype)
            self.dev.upload(_lib.VAR_CX, np.ascontiguousarray(self.Z.transpose(0, 1, 3, 2, 4)))
            self.dev.ccmod_setcoef(_lib.VAR_CX)
            self._cache.pop(_lib.VAR_CX, None)
            return
        self.Z = np.asarray(Z.reshape(self.cri.Nv + (1, self.Nb, self.cri.M)), dtype=self.dtype)
        self.dev.upload(_lib.VAR_AX, self.Z)       # staging in a free X-sized real array
        self.dev.ccmod_setcoef(_lib.VAR_AX)

    def setcoef_from_device(self, var=_lib.VAR_Y):
        """Zf = rfftn(<real state of the shared solver>), no host round trip."""
        self.dev.ccmod_setcoef(var)

    def getdict(self, crop=True):
        """The dictionary, cropped to the filter support by default (ccmod.py:331-339,
        :839-848)."""
        if crop and self._dim3:
            return cr.bcrop(self.Y, self._dsz3, 3)
        if crop:
            return self._drop1(self._from_pair(self.dev.ccmod_getdict(self.cri.mxsz[0], self.cri.mxsz[1])))
        return self.Y

    def _reconstruct_complex(self, D):
        """ifftn(sum_m fftn(Z) fftn(D)) (ccmod.py:418-427 / :897-907 on the fftn path)."""
        D = self.Y if D is None else np.asarray(D)
        Sf = np.sum(np.fft.fftn(self._Zc, axes=(0, 1)) * np.fft.fftn(D, axes=(0, 1)),
                    axis=self.cri.axisM)
        return np.fft.ifftn(Sf, axes=(0, 1)).astype(self.cdtype)

    def finish_solve(self):
        self.dev.sync()

    def rescale_u(self, rsf):
        self._u_scale = self._u_scale / float(rsf)

    def eval_objfn(self):
        return (self.obfn_dfd(), self.obfn_cns())

    def obfn_dfd(self):
        """(1/2) ||sum_m Zf Df - Sf||^2 at the variable the options select (ccmod.py:396-403,
        :871-878)."""
        return self._sums[_lib.OUT_DFID] / 2.0

    def obfn_cns(self):
        """||Pcn(v) - v||_2 (ccmod.py:407-410, :882-889)."""
        return np.sqrt(self._sums[_lib.OUT_CNSTR])

    def itstat_extra(self):
        return (self.xrrs,)

    def profile(self, enable=True):
        self.dev.profile(enable)

    def profile_read(self):
        return self.dev.profile_read()


class ConvCnstrMOD_Consensus(_DeviceDStep, admm.ADMM):
    r"""Minimise (1/2) sum_k ||sum_m d_m * x_{k,m} - s_k||_2^2 over filters d_m of unit norm
    and constrained support, as an ADMM consensus problem over the images.

    IterationStats fields: ``Iter, DFid, Cnstr, PrimalRsdl, DualRsdl, EpsPrimal, EpsDual,
    Rho, XSlvRelRes, Time``.
    """

    class Options(admm.ADMM.Options):
        """Options of sporco/admm/ccmod.py:623-648: those of ConvCnstrMODBase
        (``AuxVarObj, fEvalX, gEvalY, ReturnX, ZeroMean, LinSolveCheck``; :129-136) overlaid with
        the ADMMConsensus defaults (admm.py:1495-1497), i.e. the objective is evaluated at the
        consensus variable and AutoRho is off unless enabled; ``RelaxParam`` 1.8."""

        defaults = copy.deepcopy(admm.ADMM.Options.defaults)
        defaults.update({'AuxVarObj': True, 'fEvalX': False, 'gEvalY': True, 'ReturnX': False,
                         'ZeroMean': False, 'LinSolveCheck': False, 'RelaxParam': 1.8})

        def __init__(self, opt=None):
            admm.ADMM.Options.__init__(self, {} if opt is None else opt)
            if self['AutoRho', 'RsdlTarget'] is None:
                self['AutoRho', 'RsdlTarget'] = 1.0

        def __setitem__(self, key, value):
            admm.ADMM.Options.__setitem__(self, key, value)
            if key == 'AuxVarObj':
                self['fEvalX'] = value is not True
                self['gEvalY'] = value is True

    # sums of one iteration that are rank-local when the images are sharded (the others are
    # dictionary sized, i.e. identical on every rank)
    _shard_slots = (_lib.OUT_R2, _lib.OUT_AX2, _lib.OUT_U2, _lib.OUT_DFID)
    _mask_dcpl = False
    _objective_at_y_only = True      # (of the complex-valued form: _DeviceDStep._complex_setup)

    def __init__(self, Z, S, dsz, opt=None, dimK=1, dimN=2, device=0, stream=None, dev=None,
                 reducer=None):
        """``Z, S, dsz, opt, dimK, dimN`` as in the reference (ccmod.py:653-712).  Backend
        keywords: ``dev``, a :class:`sporco_amd._lib.Solver` to share with the sparse coding
        step, so that coefficient maps and dictionary stay on the GPU; ``reducer``
        (:class:`sporco_amd.dist.TorchReducer`), one process per GPU with ``S`` / ``Z`` holding
        this rank's images: the consensus average over the images -- the ystep of
        admm.py:1585-1591 -- becomes one all-reduce of a dictionary-sized array per iteration
        (SURVEY.md 8(e)), the X-sized sums are added over the ranks, and every rank holds
        the same dictionary."""
        self._reducer = reducer
        if opt is None:
            opt = ConvCnstrMOD_Consensus.Options()
        if not self._mask_dcpl:
            S, dsz, opt, dimN = self._signal_setup(S, dsz, opt, dimN)
            S, dsz, opt, dimK, dimN = self._volume_setup(S, dsz, opt, dimK, dimN, reducer)
        if dimN != 2:
            raise NotImplementedError("sporco_amd handles dimN = 2 (images) and, without a mask, "
                                      "dimN = 1 (signals) and -- the consensus update -- 3 (volumes)")
        if self._mask_dcpl and (np.iscomplexobj(S) or (Z is not None and np.iscomplexobj(Z))):
            raise NotImplementedError("complex-valued dictionary update: not with mask decoupling")
        Z, S, dsz, dimK = self._complex_setup(Z, S, dsz, opt, dimK, dimN, dev, reducer)
        self.cri = cr.CDU_ConvRepIndexing(dsz, S, dimK=dimK, dimN=dimN)
        if self._dim1 and self.cri.Cd > 1 and opt['ZeroMean']:
            raise NotImplementedError("dimN = 1 with a multi-channel dictionary and ZeroMean "
                                      "(see pgm/ccmod.py)")
        if self.cri.Cd > 1 and reducer is not None:
            raise NotImplementedError("a multi-channel dictionary in the consensus update: not "
                                      "with image shards")
        if (not opt['gEvalY'] or opt['fEvalX']) and (reducer is not None or self._mask_dcpl):
            raise NotImplementedError("objective at the blocks X_n (AuxVarObj False): not with "
                                      "image shards (the mean of X runs over ALL images) and not "
                                      "with mask decoupling")
        if opt['LinSolveCheck'] and reducer is not None:
            raise NotImplementedError("LinSolveCheck of the consensus D-step: not with image "
                                      "shards (its residual is of sums over ALL images)")
        self.set_dtype(opt, S.dtype)
        if self.dtype not in (np.float32, np.float64):
            raise TypeError("sporco_amd works in float32 or float64, not %s" % self.dtype)
        self._attach_device(S, dev, device, stream)     # one consensus block per image
        # (the complex form counts complex elements: Nx, Nc and the shapes are its own)
        self.yshape = self.cri_c.shpD if self._cplx else self.cri.shpD
        if self._dim1:
            self.yshape = self.yshape[1:]
        if self._dim3:
            self.yshape = self._cri3.shpD
        self.xshape = self.yshape + (self.Nb,)
        self._crop = self._dsz3[1:3] if self._dim3 else tuple(self.cri.mxsz[0:2])
        # (blocks of the whole problem: the residual scalings and tolerances refer to them)
        from ..dist import global_count
        self._nb_all = global_count(reducer, self.Nb)
        Nx = self._nb_all * int(np.prod(self.yshape))
        super(ConvCnstrMOD_Consensus, self).__init__(Nx, self.yshape, self.xshape, S.dtype, opt)
        self.Nc = self._nb_all * int(np.prod(self.yshape))
        # (the reference's `dval=cri.K` at ccmod.py:700 never takes effect: the base class has
        # already set rho, default 1.0)
        self.xrrs = None
        if Z is not None:
            self.setcoef(Z)

    # -- state ---------------------------------------------------------------------------
    def init_state(self, yshape, ushape):
        """Y = Y0 or zeros, U_n = Y0 / rho or zeros (admm.py:262-272 with uinit of
        ccmod.py:734-742), on the device."""
        self.dev.cns_init(self._to_pair(self.opt['Y0']), float(self.rho))
        if self.opt['U0'] is not None:
            self.U = self.opt['U0']

    def _from_blocks(self, a):
        """Device layout (H, W, C, K, M) -> the reference's (H, W, 1, 1, M, Nb): the channels of
        a multi-channel signal count as images (Nb = C K, channel-major: ccmod.py:757-765).
        Multi-channel dictionary: device (H, W, K, Cd, M) -> (H, W, Cd, 1, M, Nb = K)."""
        if self.cri.Cd > 1:
            return np.ascontiguousarray(np.moveaxis(a, 2, -1)[:, :, :, np.newaxis])
        a = a.reshape(a.shape[0], a.shape[1], 1, -1, a.shape[-1])
        return np.ascontiguousarray(np.moveaxis(a, 3, -1)[:, :, :, np.newaxis])

    def _to_blocks(self, a):
        if self.cri.Cd > 1:
            return np.ascontiguousarray(np.moveaxis(np.asarray(a)[:, :, :, 0], -1, 2))
        a = np.moveaxis(np.asarray(a)[:, :, :, 0], -1, 3)          # (H, W, 1, Nb, M)
        return np.ascontiguousarray(a.reshape(a.shape[0], a.shape[1], self.cri.C, -1, a.shape[-1]))

    @property
    def X(self):
        return self._drop1(self._from_pair(self._from_blocks(self.dev.download(_lib.VAR_CX))))

    @X.setter
    def X(self, value):
        if value is not None:
            self.dev.upload(_lib.VAR_CX, self._to_blocks(np.asarray(self._to_pair(self._add1(value, 6)),
                                                                    dtype=self.dtype)))

    @property
    def U(self):
        u = self._from_blocks(self.dev.download(_lib.VAR_CU))
        return self._drop1(self._from_pair(u * u.dtype.type(self._u_scale) if self._u_scale != 1.0 else u))

    @U.setter
    def U(self, value):
        if value is not None:
            self.dev.upload(_lib.VAR_CU, self._to_blocks(np.asarray(self._to_pair(self._add1(value, 6)),
                                                                    dtype=self.dtype)))
            self._u_scale = 1.0

    # -- iteration --------------------------------------------------------------------------
    def iteration(self):
        admm.refuse_step_overrides(self)
        flags = 0
        if self._needs_residuals():
            flags |= _lib.FLAG_RESID
        if not self.opt['FastSolve']:
            flags |= _lib.FLAG_OBJ
        if not self.opt['fEvalX']:
            flags |= _lib.FLAG_FEVAL_Y
        if self.opt['gEvalY']:
            flags |= _lib.FLAG_GEVAL_Y
        if self.opt['LinSolveCheck']:
            flags |= _lib.FLAG_XRRS
        self._sums = self._device_iteration(flags)
        self._u_scale = 1.0
        self._cache.clear()
        if self.opt['LinSolveCheck']:
            # rrs(sum_n ax_n, sum_n b_n) (ccmod.py:783-792; linalg.rrs, linalg.py:1126-1153)
            s = self._sums
            # (the mask-decoupled call returns block-1 sums in the XRRS slots: csc_api.hip cns_md_iter)
            d2, a2, b2 = (_lib.OUT_L1, _lib.OUT_RGR, _lib.OUT_CGN) if self._mask_dcpl else \
                (_lib.OUT_XRRS_D2, _lib.OUT_XRRS_AX2, _lib.OUT_XRRS_B2)
            nrm = max(np.sqrt(s[a2]), np.sqrt(s[b2]))
            self.xrrs = np.sqrt(s[d2]) / nrm if nrm > 0.0 else 0.0
        if not self._needs_residuals():
            return None
        self.timer.stop('solve_wo_rsdl')
        res = self.compute_residuals()
        self.timer.start('solve_wo_rsdl')
        return res

    def _device_iteration(self, flags):
        """One ``sporco_amd_csc_cns_iter`` call -- or, with image shards, its two phases around
        the all-reduce that turns the rank-local mean into the consensus average."""
        args = (self.rho, self.rlx, self._u_scale, flags, self._crop[0], self._crop[1],
                self.opt['ZeroMean'])
        if self._reducer is None:
            return self.dev.cns_iter(*args, mask_dcpl=self._mask_dcpl)
        self.dev.cns_iter(*args, mask_dcpl=self._mask_dcpl, phase=1)
        ptr, count = self.dev.cns_mean_ptr()
        self._reducer.all_reduce_ptr(self.dev, ptr, count, self.dtype == np.float32,
                                     prescale=float(self.Nb) / float(self._nb_all))
        sums = self.dev.cns_iter(*args, mask_dcpl=self._mask_dcpl, phase=2)
        return self._reducer.sum_slots(sums, self._shard_slots)

    def residual_norms(self):
        """Consensus residuals and normalisations (admm.py:1673-1707)."""
        s = self._sums
        rho, nb = float(self.rho), float(self._nb_all)
        nr = np.sqrt(s[_lib.OUT_R2])
        ns = np.sqrt(nb) * rho * np.sqrt(s[_lib.OUT_S2])
        rn = max(np.sqrt(s[_lib.OUT_AX2]), np.sqrt(nb) * np.sqrt(s[_lib.OUT_Y2]))
        sn = rho * np.sqrt(s[_lib.OUT_U2])
        return nr, ns, rn, sn

    def reconstruct(self, D=None):
        """irfftn(sum_m Zf * Df) (ccmod.py:897-907); host arithmetic, off the iteration path."""
        if self._dim3:
            raise NotImplementedError("reconstruct() of the consensus update: dimN <= 2")
        if self._cplx:
            return self._reconstruct_complex(D)
        Df = self.dev.download(_lib.VAR_DXF) if D is None else \
            np.fft.rfftn(self._add1(D, 5), axes=(0, 1))
        Zf = np.fft.rfftn(self.Z, axes=(0, 1)) if getattr(self, '_z_chan', False) else \
            self.dev.download(_lib.VAR_ZF)
        return self._drop1(np.fft.irfftn(np.sum(Zf * Df, axis=self.cri.axisM), self.cri.Nv,
                                         axes=(0, 1)).astype(self.dtype))


class ConvCnstrMODBase(_DeviceDStep, admm.ADMM):
    r"""Shared part of the single-copy ADMM dictionary updates (ConvCnstrMODBase,
    sporco/admm/ccmod.py:103-429, on ADMMEqual): X, Y and U are all one zero-padded dictionary
    (H, W, 1, 1, M); one iteration is one call of ``sporco_amd_csc_dstep_iter``.

    State, iteration and residuals are those of ADMMEqual (admm.py:808-983); the parts that do
    not depend on the splitting come from :class:`_DeviceDStep`.
    """

    class Options(admm.ADMM.Options):
        """ConvCnstrMODBase.Options (ccmod.py:109-177): ADMMEqual options plus ``AuxVarObj,
        ZeroMean, LinSolveCheck``; AutoRho on with period 1."""

        defaults = copy.deepcopy(admm.ADMM.Options.defaults)
        defaults.update({'AuxVarObj': False, 'fEvalX': True, 'gEvalY': False, 'ReturnX': False,
                         'RelaxParam': 1.8, 'ZeroMean': False, 'LinSolveCheck': False})
        defaults['AutoRho'].update({'Enabled': True, 'Period': 1, 'AutoScaling': True,
                                    'Scaling': 1000.0, 'RsdlRatio': 1.2})

        def __init__(self, opt=None):
            admm.ADMM.Options.__init__(self, {} if opt is None else opt)
            if self['AutoRho', 'RsdlTarget'] is None:
                self['AutoRho', 'RsdlTarget'] = 1.0

        def __setitem__(self, key, value):
            admm.ADMM.Options.__setitem__(self, key, value)
            if key == 'AuxVarObj':
                self['fEvalX'] = value is not True
                self['gEvalY'] = value is True

    _method = None
    _objective_at_y_only = False
    _signals_ok = True        # (the mask-decoupled subclasses: dimN = 2 only)

    def __init__(self, Z, S, dsz, opt=None, dimK=1, dimN=2, device=0, stream=None, dev=None):
        if opt is None:
            opt = type(self).Options()
        if type(self)._signals_ok:
            S, dsz, opt, dimN = self._signal_setup(S, dsz, opt, dimN)
        if dimN != 2:
            raise NotImplementedError("sporco_amd handles dimN = 2 (images) and, without a mask, "
                                      "dimN = 1 (signals)")
        Z, S, dsz, dimK = self._complex_setup(Z, S, dsz, opt, dimK, dimN, dev)
        self.cri = cr.CDU_ConvRepIndexing(dsz, S, dimK=dimK, dimN=dimN)
        if self._dim1 and self.cri.Cd > 1 and opt['ZeroMean']:
            raise NotImplementedError("dimN = 1 with a multi-channel dictionary and ZeroMean "
                                      "(see pgm/ccmod.py)")
        if opt['ReturnX']:
            raise NotImplementedError("the device D-step returns the constrained variable Y "
                                      "(ReturnX False, the class default)")
        self.set_dtype(opt, S.dtype)
        if self.dtype not in (np.float32, np.float64):
            raise TypeError("sporco_amd works in float32 or float64, not %s" % self.dtype)
        self.Nb = self.cri.C * self.cri.K
        self._attach_device(S, dev, device, stream)
        shpD = self.cri_c.shpD if self._cplx else self.cri.shpD    # (complex elements counted once)
        if self._dim1:
            shpD = shpD[1:]
        Nx = int(np.prod(shpD))
        admm.ADMM.__init__(self, Nx, shpD, shpD, S.dtype, opt)
        # (as for the consensus class, the `dval=cri.K` of ccmod.py:264 never takes effect)
        self.xrrs = None
        self.cgit = None
        if Z is not None:
            self.setcoef(Z)

    # -- state ---------------------------------------------------------------------------
    def init_state(self, yshape, ushape):
        """Y = Y0 or zeros, U = Y0 or zeros (uinit, ccmod.py:298-307), Xf = 0."""
        self.dev.dstep_init(self._to_pair(self.opt['Y0']))
        if self.opt['U0'] is not None:
            self.U = self.opt['U0']

    @property
    def X(self):
        return self._drop1(self._from_pair(self.dev.download(_lib.VAR_DSX)))

    @X.setter
    def X(self, value):
        if value is not None:
            self.dev.upload(_lib.VAR_DSX, np.asarray(self._to_pair(self._add1(value, 5)), dtype=self.dtype))

    @property
    def Xf(self):
        if self._cplx:      # (the device holds the paired spectra: the plain transform of X instead)
            return np.fft.fftn(self.X, axes=(0, 1))
        return self._drop1(self.dev.download(_lib.VAR_DYF))

    @property
    def U(self):
        u = self.dev.download(_lib.VAR_DSU)
        return self._drop1(self._from_pair(u * u.dtype.type(self._u_scale) if self._u_scale != 1.0 else u))

    @U.setter
    def U(self, value):
        if value is not None:
            self.dev.upload(_lib.VAR_DSU, np.asarray(self._to_pair(self._add1(value, 5)), dtype=self.dtype))
            self._u_scale = 1.0

    # -- iteration --------------------------------------------------------------------------
    def _cg_options(self):
        return 1e-3, 1000

    def iteration(self):
        admm.refuse_step_overrides(self)
        flags = 0
        if not self.opt['FastSolve']:
            flags |= _lib.FLAG_OBJ
        if not self.opt['fEvalX']:
            flags |= _lib.FLAG_FEVAL_Y
        if self.opt['gEvalY']:
            flags |= _lib.FLAG_GEVAL_Y
        if self.opt['LinSolveCheck']:
            flags |= _lib.FLAG_XRRS
        tol, mit = self._cg_options()
        s = self._sums = self.dev.dstep_iter(
            self._method, self.rho, self.rlx, self._u_scale, flags, self.cri.mxsz[0], self.cri.mxsz[1], self.opt['ZeroMean'], tol, mit)
        self._u_scale = 1.0
        self._cache.clear()
        if self.opt['LinSolveCheck']:
            nrm = max(np.sqrt(s[_lib.OUT_XRRS_AX2]), np.sqrt(s[_lib.OUT_XRRS_B2]))
            self.xrrs = np.sqrt(s[_lib.OUT_XRRS_D2]) / nrm if nrm > 0.0 else 0.0
        if self._method == _lib.DSTEP_CG:
            self.cgit = int(s[_lib.OUT_CGIT])
            self.cg_iterations = int(s[_lib.OUT_CGN])
        if not self._needs_residuals():
            return None
        self.timer.stop('solve_wo_rsdl')
        res = self.compute_residuals()
        self.timer.start('solve_wo_rsdl')
        return res

    def residual_norms(self):
        """ADMMEqual residuals and normalisations (admm.py:959-983)."""
        s = self._sums
        rho = float(self.rho)
        return (np.sqrt(s[_lib.OUT_R2]), rho * np.sqrt(s[_lib.OUT_S2]),
                max(np.sqrt(s[_lib.OUT_AX2]), np.sqrt(s[_lib.OUT_Y2])),
                rho * np.sqrt(s[_lib.OUT_U2]))

    def reconstruct(self, D=None):
        """irfftn(sum_m Zf * Xf) (ccmod.py:418-427); host arithmetic, off the iteration path."""
        if self._cplx:
            return self._reconstruct_complex(self.X if D is None else D)
        Df = self.dev.download(_lib.VAR_DYF) if D is None else np.fft.rfftn(self._add1(D, 5), axes=(0, 1))
        Zf = self.dev.download(_lib.VAR_ZF)
        return self._drop1(np.fft.irfftn(np.sum(Zf * Df, axis=self.cri.axisM), self.cri.Nv,
                                         axes=(0, 1)).astype(self.dtype))


class ConvCnstrMOD_IterSM(ConvCnstrMODBase):
    r"""ADMM dictionary update with the X-step solved by iterated Sherman-Morrison over the
    images (sporco/admm/ccmod.py:433-505; linalg.solvemdbi_ism).  Up to 8 images (times
    channels) the rank-one terms of one frequency live in registers; beyond that the same
    recursion re-reads them from memory (the cost grows with the square of the number of
    images, which is why the reference recommends this method for small training sets).

    IterationStats fields: ``Iter, DFid, Cnstr, PrimalRsdl, DualRsdl, EpsPrimal, EpsDual,
    Rho, XSlvRelRes, Time``.
    """

    class Options(ConvCnstrMODBase.Options):
        defaults = copy.deepcopy(ConvCnstrMODBase.Options.defaults)

    _method = _lib.DSTEP_ISM


class ConvCnstrMOD_CG(ConvCnstrMODBase):
    r"""ADMM dictionary update with the X-step solved by conjugate gradients, warm-started from
    the previous solution (sporco/admm/ccmod.py:511-601; linalg.solvemdbi_cg with scipy's cg
    stopping rule).  ``XSlvCGIt`` carries scipy's status flag as in the reference (0 when the
    tolerance was met, ``MaxIter`` otherwise); the attribute ``cg_iterations`` has the count.

    IterationStats fields: ``Iter, DFid, Cnstr, PrimalRsdl, DualRsdl, EpsPrimal, EpsDual,
    Rho, XSlvRelRes, XSlvCGIt, Time``.
    """

    class Options(ConvCnstrMODBase.Options):
        """Adds ``CG``: ``MaxIter`` (1000), ``StopTol`` (1e-3) (ccmod.py:530-545)."""
        defaults = copy.deepcopy(ConvCnstrMODBase.Options.defaults)
        defaults.update({'CG': {'MaxIter': 1000, 'StopTol': 1e-3}})

    itstat_fields_extra = ('XSlvRelRes', 'XSlvCGIt')
    _method = _lib.DSTEP_CG

    def _cg_options(self):
        return self.opt['CG', 'StopTol'], self.opt['CG', 'MaxIter']

    def itstat_extra(self):
        return (self.xrrs, self.cgit)


_METHODS = {'cns': ConvCnstrMOD_Consensus, 'ism': ConvCnstrMOD_IterSM, 'cg': ConvCnstrMOD_CG}


def _lookup(method):
    if method in _METHODS:
        return _METHODS[method]
    raise ValueError('Unknown ConvCnstrMOD solver method %s' % method)


def ConvCnstrMODOptions(opt=None, method='cns'):
    """Options object of the selected ADMM dictionary update (ccmod.py:953-990)."""
    return _lookup(method).Options(opt)


def ConvCnstrMOD(*args, **kwargs):
    """Construct the ADMM dictionary update selected by ``method`` (ccmod.py:911-950)."""
    method = kwargs.pop('method', 'cns')
    return _lookup(method)(*args, **kwargs)
