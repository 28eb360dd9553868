"""ADMM dictionary updates with mask decoupling on the GPU.

Drop-ins for ``sporco.admm.ccmodmd.ConvCnstrMODMaskDcpl_IterSM`` and ``ConvCnstrMODMaskDcpl_CG``
(sporco/admm/ccmodmd.py:573-762 on ``ConvCnstrMODMaskDcplBase`` :27-567 and
``admm.ADMMTwoBlockCnstrnt``, sporco/admm/admm.py:989-1437): minimise
(1/2) sum_k ||W (sum_m d_m * x_{k,m} - s_k)||^2 over constrained filters through the two-block
constraint [Z; I] d - [y0; y1] = [s; 0]; and for the hybrid consensus form
``ConvCnstrMODMaskDcpl_Consensus`` (:766-1083, at the end of this module).

One iteration is one call of ``sporco_amd_csc_dstep_iter`` with ``mask_dcpl`` set: the X-step
solves (Z^H Z + I) Xf = sum_n conj(Zf_n) rfftn(y0 - u0 + s)_n + rfftn(y1 - u1) with the kernels
of the unmasked updates (:mod:`sporco_amd.admm.ccmod`; rho does not enter), block 1 (dictionary
sized) follows their relax / Pcn / dual-update path and block 0 (signal sized) the small kernel
of :class:`sporco_amd.admm.cbpdn.ConvBPDNMaskDcpl`.
"""

import copy

import numpy as np

from . import admm
from . import ccmod
from .. import _lib
from .. import cnvrep as cr

__all__ = ['ConvCnstrMODMaskDcpl_IterSM', 'ConvCnstrMODMaskDcpl_CG',
           'ConvCnstrMODMaskDcpl_Consensus', 'ConvCnstrMODMaskDcpl',
           'ConvCnstrMODMaskDcplOptions']


class ConvCnstrMODMaskDcplBase(ccmod.ConvCnstrMODBase):
    r"""Shared part (ConvCnstrMODMaskDcplBase, sporco/admm/ccmodmd.py:27-567).

    IterationStats fields: ``Iter, DFid, Cnstr, PrimalRsdl, DualRsdl, EpsPrimal, EpsDual,
    Rho, XSlvRelRes, Time``.  ``Y0`` is the initial dictionary (block 1) as dictionary learning
    passes it, or the reference's concatenated [y0; y1] array whose block 0 must then be zero.
    """

    class Options(admm.ADMM.Options):
        """ccmodmd.py:137-201: ADMM base defaults (AutoRho off) with ``rho`` 1.0, ``RelaxParam``
        1.8, ``ReturnVar`` 'Y1', ``AuxVarObj``, ``LinSolveCheck``, ``ZeroMean``."""

        defaults = copy.deepcopy(admm.ADMMEqual.Options.defaults)
        defaults.update({'AuxVarObj': False, 'fEvalX': True, 'gEvalY': False,
                         'LinSolveCheck': False, 'ZeroMean': False, 'RelaxParam': 1.8,
                         'rho': 1.0, 'ReturnVar': 'Y1', 'ReturnX': False})

        def __init__(self, opt=None):
            admm.ADMM.Options.__init__(self, {} if opt is None else opt)

        def __setitem__(self, key, value):
            admm.ADMM.Options.__setitem__(self, key, value)
            if key == 'AuxVarObj':
                self['fEvalX'] = value is not True
                self['gEvalY'] = value is True

    _signals_ok = False

    def __init__(self, Z, S, W, dsz, opt=None, dimK=1, dimN=2, device=0, stream=None, dev=None):
        if opt is None:
            opt = type(self).Options()
        if opt['ReturnVar'] != 'Y1':
            raise NotImplementedError("the device D-step returns the dictionary (ReturnVar 'Y1', "
                                      "the class default)")
        cri = cr.CDU_ConvRepIndexing(dsz, S, dimK=dimK, dimN=dimN)
        Nb = (cri.C if cri.Cd == 1 else 1) * cri.K     # (CK of ccmodmd.py:273)
        y0 = opt['Y0']
        if y0 is not None and np.asarray(y0).shape[-1] == Nb + cri.M:
            # the reference's [y0; y1] on the filter axis (ccmodmd.py:400-445)
            y0 = np.asarray(y0)
            if np.any(y0[..., :Nb] != 0):
                raise NotImplementedError("a warm start of block 0 is not offered")
            opt = copy.deepcopy(opt)
            opt['Y0'] = np.ascontiguousarray(y0[..., Nb:])
        W = np.asarray(W)
        self._W_in = W
        self._S_in = np.asarray(S)
        super(ConvCnstrMODMaskDcplBase, self).__init__(Z, S, dsz, opt, dimK=dimK, dimN=dimN,
                                                       device=device, stream=stream, dev=dev)
        # problem sizes of the two-block constraint (ccmodmd.py:262-270)
        self.Nc = self.Nx + int(np.prod(self.S.shape))
        self._nrm_c = float(np.linalg.norm(self.S))

    def _mask5(self):
        """The mask in the internal layout with channels folded into the image axis
        (ccmodmd.py:243-259)."""
        cri, W = self.cri, self._W_in
        if W.size == 1:
            return np.asarray(W.reshape((1,) * 5), dtype=self.dtype)
        W = np.asarray(W.reshape(cr.mskWshape(W, cri)), dtype=self.dtype)
        if cri.Cd == 1 and cri.C > 1:
            shpw = list(W.shape)
            swck = shpw[cri.axisC] * shpw[cri.axisK]
            if 1 < swck < cri.C * cri.K:
                if W.shape[cri.axisK] == 1 and cri.K > 1:
                    shpw[cri.axisK] = cri.K
                else:
                    shpw[cri.axisC] = cri.C
                W = np.broadcast_to(W, shpw)
            W = W.reshape(W.shape[0:cri.dimN] + (1, W.shape[cri.axisC] * W.shape[cri.axisK], 1))
        return W

    def init_state(self, yshape, ushape):
        """y1 = u1 = Y0 or zeros, block 0 zero (uinit, ccmodmd.py:311-322), Xf = 0; the mask and
        the real signal go to the device."""
        self.W = self._mask5()
        H, Wd = self.cri.Nv
        if self.cri.Cd > 1:
            # multi-channel dictionary: signal, mask and block 0 keep the channel axis
            self.dev.set_data_mask(ccmod_broadcastable(
                self.W, (H, Wd, self.cri.C, self.cri.K, 1)))
        elif self.cri.C > 1:
            # the handle keeps channels and images on separate axes; (H, W, 1, C K) and
            # (H, W, C, K) share their memory layout
            full = np.ascontiguousarray(np.broadcast_to(self.W, (H, Wd, 1, self.Nb, 1)))
            self.dev.set_data_mask(full.reshape(H, Wd, self.cri.C, self.cri.K, 1))
        else:
            self.dev.set_data_mask(ccmod_broadcastable(self.W, (H, Wd, 1, self.Nb, 1)))
        self.dev.dstep_md_init(self.opt['Y0'], self.S)

    # -- the two blocks, in the reference's layout (y0 moved to the filter axis) ---------------
    def var_y0(self):
        return np.moveaxis(self.dev.download(_lib.VAR_DMY0), 3, 4)

    def var_y1(self):
        return super(ConvCnstrMODMaskDcplBase, self).Y

    @property
    def Y(self):
        return np.concatenate((self.var_y0(), self.var_y1()), axis=self.cri.axisM)

    @Y.setter
    def Y(self, value):
        if value is not None:
            raise NotImplementedError("the blocks of Y are device state")

    @property
    def U(self):
        u0 = np.moveaxis(self.dev.download(_lib.VAR_DMU0), 3, 4)
        if self._u_scale != 1.0:
            u0 = u0 * u0.dtype.type(self._u_scale)
        return np.concatenate((u0, super(ConvCnstrMODMaskDcplBase, self).U), axis=self.cri.axisM)

    @U.setter
    def U(self, value):
        if value is not None:
            raise NotImplementedError("the blocks of U are device state")

    def getmin(self):
        return self.var_y1()

    def getdict(self, crop=True):
        if crop:
            return self.dev.ccmod_getdict(self.cri.mxsz[0], self.cri.mxsz[1])
        return self.var_y1()

    # -- iteration ----------------------------------------------------------------------------
    def iteration(self):
        admm.refuse_step_overrides(self)
        flags = 0
        if not self.opt['FastSolve']:
            flags |= _lib.FLAG_OBJ
        if self.opt['AuxVarObj']:
            flags |= _lib.FLAG_GEVAL_Y
        if self.opt['LinSolveCheck']:
            flags |= _lib.FLAG_XRRS
        tol, mit = self._cg_options()
        s = self._sums = self.dev.dstep_iter(
            self._method, self.rho, self.rlx, self._u_scale, flags, self.cri.mxsz[0], self.cri.mxsz[1], self.opt['ZeroMean'], tol, mit, mask_dcpl=True)
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
        """admm.py:1404-1437 with the dual residual of ccmodmd.py:557-567."""
        s = self._sums
        rho = float(self.rho)
        nr = np.sqrt(s[_lib.OUT_R2] + s[_lib.OUT_L1])
        ns = rho * np.sqrt(s[_lib.OUT_S2])
        rn = max(np.sqrt(s[_lib.OUT_AX2] + s[_lib.OUT_L21]),
                 np.sqrt(s[_lib.OUT_Y2] + s[_lib.OUT_RGR]), self._nrm_c)
        sn = rho * np.sqrt(s[_lib.OUT_U2] + s[15])
        return nr, ns, rn, sn


def ccmod_broadcastable(w, full_shape):
    """``w`` with every axis 1 or full, as the device weight upload wants it."""
    w = np.asarray(w)
    shape = tuple(f if d == f else 1 for d, f in zip(w.shape, full_shape))
    if any(d not in (1, f) for d, f in zip(w.shape, full_shape)):
        raise ValueError("mask of shape %s does not fit data of shape %s" % (w.shape, full_shape))
    return np.ascontiguousarray(w.reshape(shape))


class ConvCnstrMODMaskDcpl_IterSM(ConvCnstrMODMaskDcplBase):
    r"""X-step by iterated Sherman-Morrison over the images (ccmodmd.py:573-654), as
    :class:`sporco_amd.admm.ccmod.ConvCnstrMOD_IterSM`."""

    class Options(ConvCnstrMODMaskDcplBase.Options):
        defaults = copy.deepcopy(ConvCnstrMODMaskDcplBase.Options.defaults)

    _method = _lib.DSTEP_ISM


class ConvCnstrMODMaskDcpl_CG(ConvCnstrMODMaskDcplBase):
    r"""X-step by warm-started conjugate gradients (ccmodmd.py:658-762); ``XSlvCGIt`` is scipy's
    status flag as in the reference."""

    class Options(ConvCnstrMODMaskDcplBase.Options):
        """Adds ``CG``: ``MaxIter`` (1000), ``StopTol`` (1e-3) (ccmodmd.py:678-709)."""
        defaults = copy.deepcopy(ConvCnstrMODMaskDcplBase.Options.defaults)
        defaults.update({'CG': {'MaxIter': 1000, 'StopTol': 1e-3}})

    itstat_fields_extra = ('XSlvRelRes', 'XSlvCGIt')
    _method = _lib.DSTEP_CG

    def _cg_options(self):
        return self.opt['CG', 'StopTol'], self.opt['CG', 'MaxIter']

    def itstat_extra(self):
        return (self.xrrs, self.cgit)


class ConvCnstrMODMaskDcpl_Consensus(ccmod.ConvCnstrMOD_Consensus):
    r"""Hybrid consensus / mask-decoupling dictionary update (sporco/admm/ccmodmd.py:766-1083):
    minimise (1/2) ||W (sum_m d_m * x_m - s)||_2^2 over filters of unit norm and constrained
    support, with one dictionary copy per image (the consensus splitting of
    :class:`sporco_amd.admm.ccmod.ConvCnstrMOD_Consensus`) and a signal-sized block ``Y1, U1``
    that carries the mask.  One iteration is one call of ``sporco_amd_csc_cns_iter`` with
    ``mask_dcpl`` set; the host forms the residuals of :976-1034 from the returned sums.

    IterationStats fields: ``Iter, DFid, Cnstr, PrimalRsdl, DualRsdl, EpsPrimal, EpsDual,
    Rho, XSlvRelRes, Time``.
    """

    _mask_dcpl = True
    _shard_slots = (_lib.OUT_R2, _lib.OUT_AX2, _lib.OUT_U2, _lib.OUT_S2, _lib.OUT_XRRS_D2,
                    _lib.OUT_XRRS_AX2, _lib.OUT_XRRS_B2, _lib.OUT_CGIT, _lib.OUT_DFID)

    def __init__(self, Z, S, W, dsz, opt=None, dimK=1, dimN=2, device=0, stream=None, dev=None,
                 reducer=None):
        """``reducer``: image shards over ranks as for
        :class:`sporco_amd.admm.ccmod.ConvCnstrMOD_Consensus` (``S``, ``W``, ``Z`` hold this
        rank's images)."""
        if opt is None:
            opt = ccmod.ConvCnstrMOD_Consensus.Options()
        self._W_in = np.asarray([1.0]) if W is None else np.asarray(W)
        super(ConvCnstrMODMaskDcpl_Consensus, self).__init__(Z, S, dsz, opt, dimK=dimK, dimN=dimN,
                                                             device=device, stream=stream, dev=dev,
                                                             reducer=reducer)
        s2 = float(np.linalg.norm(self.S)) ** 2
        if reducer is not None:
            s2 = reducer.sum([s2])[0]
        self._nrm_s = float(np.sqrt(s2))

    _mask5 = ConvCnstrMODMaskDcplBase._mask5

    def init_state(self, yshape, ushape):
        super(ConvCnstrMODMaskDcpl_Consensus, self).init_state(yshape, ushape)
        self.W = self._mask5()
        H, Wd = self.cri.Nv
        if self.cri.Cd > 1:
            # multi-channel dictionary: signal, mask and the block (Y1, U1) keep the channel axis
            self.dev.set_data_mask(ccmod_broadcastable(
                self.W, (H, Wd, self.cri.C, self.cri.K, 1)))
        elif self.cri.C > 1:
            full = np.ascontiguousarray(np.broadcast_to(self.W, (H, Wd, 1, self.Nb, 1)))
            self.dev.set_data_mask(full.reshape(H, Wd, self.cri.C, self.cri.K, 1))
        else:
            self.dev.set_data_mask(ccmod_broadcastable(self.W, (H, Wd, 1, self.Nb, 1)))
        self.dev.cns_md_init(self.S)

    # the signal-sized block, in the reference's (H, W, 1, Nb, 1) layout ((H, W, C, K, 1) with a
    # multi-channel dictionary)
    @property
    def Y1(self):
        return self.dev.download(_lib.VAR_DMY0)

    @property
    def U1(self):
        # (not touched by update_rho's rescaling of U: the reference scales self.U only)
        return self.dev.download(_lib.VAR_DMU0)

    def var_y1(self):
        """The dictionary (named for compatibility with the IterSM / CG classes, :889-897)."""
        return self.Y

    def iteration(self):
        admm.refuse_step_overrides(self)
        flags = 0
        if self._needs_residuals():
            flags |= _lib.FLAG_RESID
        if not self.opt['FastSolve']:
            flags |= _lib.FLAG_OBJ
        if self.opt['LinSolveCheck']:
            flags |= _lib.FLAG_XRRS
        self._sums = self._device_iteration(flags)
        self._u_scale = 1.0
        self._cache.clear()
        if self.opt['LinSolveCheck']:
            # rrs(sum_n ax_n, sum_n b_n) of the consensus solve (admm/ccmod.py:783-792); the XRRS
            # slots of this call carry block-1 sums, the three sums come back in L1 / RGR / CGN
            s = self._sums
            nrm = max(np.sqrt(s[_lib.OUT_RGR]), np.sqrt(s[_lib.OUT_CGN]))
            self.xrrs = np.sqrt(s[_lib.OUT_L1]) / nrm if nrm > 0.0 else 0.0
        if not self._needs_residuals():
            return None
        self.timer.stop('solve_wo_rsdl')
        res = self.compute_residuals()
        self.timer.start('solve_wo_rsdl')
        return res

    def residual_norms(self):
        """ccmodmd.py:976-1012: both blocks enter every norm; the dual residual is rho ||A^T u||
        with the new duals; the primal normalisation also sees ||s||."""
        s = self._sums
        rho = float(self.rho)
        nr = np.sqrt(s[_lib.OUT_R2] + s[_lib.OUT_XRRS_D2])
        ns = rho * np.sqrt(s[_lib.OUT_S2])
        rn = max(np.sqrt(s[_lib.OUT_AX2] + s[_lib.OUT_XRRS_AX2]),
                 np.sqrt(s[_lib.OUT_Y2] + s[_lib.OUT_XRRS_B2]), self._nrm_s)
        sn = rho * np.sqrt(s[_lib.OUT_U2] + s[_lib.OUT_CGIT])
        return nr, ns, rn, sn


_METHODS = {'ism': ConvCnstrMODMaskDcpl_IterSM, 'cg': ConvCnstrMODMaskDcpl_CG,
            'cns': ConvCnstrMODMaskDcpl_Consensus}


def _lookup(method):
    if method in _METHODS:
        return _METHODS[method]
    raise ValueError('Unknown ConvCnstrMODMaskDcpl solver method %s' % method)


def ConvCnstrMODMaskDcplOptions(opt=None, method='cns'):
    """Options object of the selected update (ccmodmd.py:1099-1132)."""
    return _lookup(method).Options(opt)


def ConvCnstrMODMaskDcpl(*args, **kwargs):
    """Construct the update selected by ``method`` (ccmodmd.py:1056-1095)."""
    method = kwargs.pop('method', 'cns')
    return _lookup(method)(*args, **kwargs)
