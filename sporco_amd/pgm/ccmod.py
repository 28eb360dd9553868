"""Constrained convolutional dictionary update by FISTA, on the GPU.

Drop-in for ``sporco.pgm.ccmod.ConvCnstrMOD`` (sporco/pgm/ccmod.py:28-404):
minimise (1/2) sum_k ||sum_m d_m * x_{k,m} - s_k||^2 over filters d_m of unit
norm and support ``dsz``.  The PGM variable is the zero-padded dictionary;
IterationStats fields are ``Iter, DFid, Cnstr, Rsdl, F_Btrack, Q_Btrack,
IterBTrack, L, Time``.

Device work per step: the gradient needs two contractions of the coefficient
spectra Zf (over filters, then over images); one kernel does both per pixel
while the slice of Zf is cache resident.  The proximal map is the constraint
projection ``Pcn`` (crop, zero-pad, optional zero-mean, normalise).
"""

import copy

import numpy as np

from . import pgm
from .. import _lib
from .. import cnvrep as cr
from ..admm.cbpdn import _DeviceArray

__all__ = ['ConvCnstrMOD', 'ConvCnstrMODMask']


def _dsz_unit_axis(dsz):
    """Filter support(s) of a dimN = 1 dictionary with a unit first axis added: (w, [C,] M) ->
    (1, w, [C,] M), for each scale of a multi-scale specification (cnvrep.py:201-215)."""
    if isinstance(dsz[0], (list, tuple)):
        return tuple(_dsz_unit_axis(d) for d in dsz)
    return (1,) + tuple(dsz)


class ConvCnstrMOD(pgm.PGMDFT):

    class Options(pgm.PGMDFT.Options):
        """Adds ``ZeroMean`` (sporco/pgm/ccmod.py:111-112)."""

        defaults = copy.deepcopy(pgm.PGMDFT.Options.defaults)
        defaults.update({'ZeroMean': False})

        def __init__(self, opt=None):
            pgm.PGMDFT.Options.__init__(self, {} if opt is None else opt)

    itstat_fields_objfn = ('DFid', 'Cnstr')
    hdrtxt_objfn = ('DFid', 'Cnstr')
    hdrval_objfun = {'DFid': 'DFid', 'Cnstr': 'Cnstr'}
    _volumes_ok = True      # (dimN = 3; the masked subclass: dimN <= 2)

    _v = {'x': _lib.VAR_DXF, 'y': _lib.VAR_DYF, 'xprv': _lib.VAR_DXFPRV,
          'yprv': _lib.VAR_DYFPRV, 'g': _lib.VAR_DGF, 't0': _lib.VAR_DT0,
          't1': _lib.VAR_DT1, 't2': _lib.VAR_DT2}

    X = _DeviceArray(_lib.VAR_DX)
    Xf = _DeviceArray(_lib.VAR_DXF)
    Yf = _DeviceArray(_lib.VAR_DYF)
    Zf = _DeviceArray(_lib.VAR_ZF)
    Sf = _DeviceArray(_lib.VAR_SF)

    def __init__(self, Z, S, dsz, opt=None, dimK=1, dimN=2, device=0, stream=None, dev=None,
                 reducer=None, _dim1=False):
        """``Z, S, dsz, opt, dimK, dimN`` as in the reference (pgm/ccmod.py:139); ``dimN`` 2
        (images) or 1 (signals: run as images with a unit first axis, which the public arrays
        -- X, getdict(), reconstruct() -- do not show).
        Backend keyword ``dev``: an existing :class:`sporco_amd._lib.Solver`
        (that of the sparse-coding step) to share, so that coefficient maps and
        dictionary never leave the GPU during dictionary learning.  ``reducer``
        (:class:`sporco_amd.dist.TorchReducer`): ``S`` / ``Z`` are this rank's block of the
        images; the gradient and the data-fidelity sums are all-reduced, the dictionary and
        its projection are replicated."""
        self._reducer = reducer
        if opt is None:
            opt = ConvCnstrMOD.Options()
        self._dim1 = bool(_dim1)
        if dimN == 1:
            self._dim1 = True
            S, dsz, dimN = np.asarray(S)[np.newaxis], _dsz_unit_axis(dsz), 2
        # dimN = 3 (volumes): the first two axes folded, on a volume handle whose projections crop in
        # three axes (admm.cbpdn.GenericConvBPDN; include/sporco_amd.h sporco_amd_csc_create_volume)
        self._dim3 = None
        if dimN == 3 and type(self)._volumes_ok:
            if isinstance(dsz[0], (list, tuple)) or reducer is not None:
                raise NotImplementedError("dimN = 3: one filter support, no image shards")
            S = np.asarray(S)
            c3 = cr.CDU_ConvRepIndexing(dsz, S, dimK=dimK, dimN=3)
            if c3.Cd > 1:
                raise NotImplementedError("dimN = 3: single-channel dictionary")
            if opt['ZeroMean']:
                # (the reference's padded projection passes no dimN to its mean subtraction,
                # cnvrep.py:1011: at dimN = 3 it takes means over the first two axes only, unlike its
                # own cropped form -- as for dimN = 1 with channels, neither reading is offered)
                raise NotImplementedError("dimN = 3 with ZeroMean")
            self._dim3 = (int(c3.Nv[0]), int(c3.Nv[1]))
            self._dsz3, self._cri3 = tuple(int(v) for v in dsz), c3
            S = cr.fold3(S.reshape(c3.shpS), *self._dim3)[..., 0]                 # (depth * H, W, C, K)
            dsz, dimK, dimN = (self._dim3[0] * self._dim3[1], int(c3.Nv[2]), c3.M), 1, 2
        if dimN != 2:
            raise NotImplementedError("sporco_amd handles dimN = 2 (images), 1 (signals) and, for the "
                                      "plain update, 3 (volumes)")
        self.cri = cr.CDU_ConvRepIndexing(dsz, S, dimK=dimK, dimN=dimN)
        # (support handed to the device's projections: rows x columns -- of one depth slab on a
        # volume handle, which knows the depth of the support by itself)
        self._crop = self._dsz3[1:3] if self._dim3 else tuple(self.cri.mxsz[0:2])
        if self._dim1 and self.cri.Cd > 1 and opt['ZeroMean']:
            # the reference's projection passes no dimN to its mean subtraction there
            # (cnvrep.py:1011 against :1074): with dimN = 1 it takes the mean over samples AND
            # channels, unlike its own cropped form; neither reading is offered as "the" result
            raise NotImplementedError("dimN = 1 with a multi-channel dictionary and ZeroMean")
        self.set_dtype(opt, S.dtype)
        if self.dtype not in (np.float32, np.float64):
            raise TypeError("sporco_amd works in float32 or float64, not %s" % self.dtype)
        self._cache = {}
        self._fcache = {}
        self._shared = dev is not None
        H, W = self.cri.Nv
        # channels of S fold into the image axis for a single-channel dictionary
        # (pgm/ccmod.py:224-229); (H, W, C, K) and (H, W, 1, C*K) share their memory layout.
        # With a multi-channel dictionary S keeps its channel axis and the coefficient maps
        # have one channel.
        if self.cri.Cd == 1:
            self.S = np.asarray(S.reshape(self.cri.Nv + (1, self.cri.C * self.cri.K, 1)),
                                dtype=self.dtype)
        else:
            self.S = np.asarray(S.reshape(self.cri.shpS), dtype=self.dtype)
        if dev is None:
            self.dev = _lib.Solver(H, W, self.cri.C, self.cri.K, self.cri.M, self.dtype,
                                   device=device, stream=stream, Cd=self.cri.Cd,
                                   depth=self._dim3[0] if self._dim3 else 1)
            self.dev.set_signal(self.S)
        else:
            if (dev.dims[:2] + (dev.Cs,) + dev.dims[3:]) != (H, W, self.cri.C, self.cri.K,
                                                              self.cri.M) \
                    or dev.Cd != self.cri.Cd or dev.dtype != self.dtype \
                    or getattr(dev, 'depth', 1) != (self._dim3[0] if self._dim3 else 1):
                raise ValueError("shared device solver has different dimensions")
            self.dev = dev
        if self._dim3:
            self.dev.set_hint(_lib.VOLUME_FILTER_DEPTH, self._dsz3[0])
        else:
            # (multi-scale dsz: every filter's own support for the projections of this handle)
            self.dev.set_filter_sizes(self.cri.fsz)
        super(ConvCnstrMOD, self).__init__(self.cri.shpD, self.cri.Nv, self.cri.axisN,
                                           S.dtype, opt)
        from ..dist import global_count
        nimg = global_count(reducer, self.cri.K)
        self.set_attr('L', opt['L'], dval=nimg * 14.0, dtype=self.dtype)
        if self._dim3:
            self.Pcn = cr.getPcn(self._dsz3, self._cri3.Nv, 3, self._cri3.dimCd, zm=opt['ZeroMean'])
        else:
            pcn = cr.getPcn(dsz, self.cri.Nv, self.cri.dimN, self.cri.dimCd, zm=opt['ZeroMean'])
            self.Pcn = (lambda x: pcn(np.asarray(x)[np.newaxis])[0]) if self._dim1 else pcn
        if Z is not None:
            self.setcoef(Z)

    # -- state ---------------------------------------------------------------------
    def _fetch(self, var):
        if var not in self._cache:
            a = self.dev.download(var)
            if self._dim1 and a.ndim >= 2 and a.shape[0] == 1:
                a = a[0]
            if self._dim3:
                a = cr.unfold3(a, *self._dim3)
            self._cache[var] = a
        return self._cache[var]

    def _store(self, var, value):
        if value is None:
            return
        value = np.asarray(value)
        if self._dim1 and value.ndim == 4:
            value = value[np.newaxis]
        if self._dim3 and value.ndim == 6:
            value = cr.fold3(value, *self._dim3)
        self.dev.upload(var, value)
        self.invalidate(var)

    def init_state(self, xshape):
        """X = X0 or zeros; Xf = rfftn(X); Yf = Xf (pgm/ccmod.py:241-252)."""
        if self.opt['X0'] is None:
            self.X = np.zeros(xshape, dtype=self.dtype)
        else:
            self.X = np.asarray(self.opt['X0']).astype(self.dtype, copy=True)
        self.dev.fft_var(_lib.VAR_DX, _lib.VAR_DXF)
        self.dev.copy(_lib.VAR_DYF, _lib.VAR_DXF)
        self.invalidate(_lib.VAR_DXF, _lib.VAR_DYF)
        self.Y = None

    def setcoef(self, Z):
        """Set the coefficient maps: Zf = rfftn(Z) (pgm/ccmod.py:264-279)."""
        self.Z = np.asarray(Z, dtype=self.dtype)
        if self._dim1 and self.Z.ndim == 4:
            self.Z = self.Z[np.newaxis]
        if self._dim3 and self.Z.shape[0:2] == self._dim3:
            self.Z = cr.fold3(self.Z.reshape(self._cri3.shpX), *self._dim3)
        cri = self.cri
        if cri.Cd > 1 and self.Z.size == cri.N * cri.Cd * cri.K * cri.M:
            # maps that carry the dictionary's channels (the reference's broadcasting makes
            # that Cd single-channel problems, tests/pgm/test_ccmod.py:175-191): handed over in
            # the (H, W, K, Cd, M) layout of the device's per-image dictionary blocks
            Zc = self.Z.reshape(cri.Nv + (cri.Cd, cri.K, cri.M)).transpose(0, 1, 3, 2, 4)
            self.dev.upload(_lib.VAR_CX, np.ascontiguousarray(Zc))
            self.dev.ccmod_setcoef(_lib.VAR_CX)
        else:
            self.dev.upload(_lib.VAR_AX, self.Z)   # staging in a free X-sized real array
            self.dev.ccmod_setcoef(_lib.VAR_AX)
        self.invalidate(_lib.VAR_ZF)
        self._fcache.clear()

    def setcoef_from_device(self, var=_lib.VAR_Y):
        """Zf = rfftn(<real state of the shared solver>), no host round trip."""
        self.dev.ccmod_setcoef(var)
        self.invalidate(_lib.VAR_ZF)
        self._fcache.clear()

    def getdict(self, crop=True):
        """Current dictionary, cropped to the filter support by default
        (pgm/ccmod.py:283-291)."""
        if crop and self._dim3:
            return cr.bcrop(self.X, self._dsz3, 3)
        if crop:
            D = self.dev.ccmod_getdict(self.cri.mxsz[0], self.cri.mxsz[1])
            return D[0] if self._dim1 else D
        return self.X

    # -- smooth term -------------------------------------------------------------------
    def grad_f(self, V=None):
        """sum_k conj(Zf) (sum_m Zf V - Sf) at V (default Yf) (pgm/ccmod.py:295-309)."""
        if V is None:
            V = _lib.VAR_DYF
        out = self.dev.ccmod_grad(V)
        if self._reducer is not None:
            self._reducer.all_reduce_array(self.dev, _lib.VAR_DGF)
            out = self._reducer.sum(out)
        self._fcache[V] = out[_lib.PGM_F]
        self.invalidate(_lib.VAR_DGF)
        return _lib.VAR_DGF

    def _eval(self, var):
        """Data-fidelity sums at a dictionary spectrum, over all ranks' images."""
        out = self.dev.ccmod_eval(var)
        return out if self._reducer is None else self._reducer.sum(out)

    def obfn_f(self, Xf=None):
        if Xf is None:
            Xf = _lib.VAR_DXF
        if Xf not in self._fcache:
            self._fcache[Xf] = self._eval(Xf)[_lib.PGM_F]
        return self._fcache[Xf]

    def hess_quad(self, V):
        return self._eval(V)[_lib.PGM_HESS]

    def prox_step(self, gradf):
        if gradf != _lib.VAR_DGF:
            self.dev.copy(_lib.VAR_DGF, gradf)
        self.dev.ccmod_prox_step(self.L, self._crop[0], self._crop[1], self.opt['ZeroMean'])
        self.invalidate(_lib.VAR_DX, _lib.VAR_DXF, _lib.VAR_DVF)

    # -- objective ------------------------------------------------------------------------
    def eval_objfn(self):
        return (self.obfn_dfd(), self.obfn_cns())

    def obfn_dfd(self):
        return self._eval(_lib.VAR_DXF)[_lib.PGM_DFID] / 2.0

    def obfn_cns(self):
        """||Pcn(X) - X||_2 (pgm/ccmod.py:350-355)."""
        return self.dev.ccmod_cnstr(self._crop[0], self._crop[1], self.opt['ZeroMean'])

    def reconstruct(self, D=None):
        """irfftn(sum_m Zf * Df) (pgm/ccmod.py:374-383); host arithmetic on the
        downloaded spectra (off the iteration path)."""
        if self._dim3:
            Nv = self._cri3.Nv
            Df = self.Xf if D is None else np.fft.rfftn(np.asarray(D), Nv, axes=(0, 1, 2))
            Sf = np.sum(self.Zf * Df, axis=5)
            return np.fft.irfftn(Sf, Nv, axes=(0, 1, 2)).astype(self.dtype)
        if self._dim1:
            Df = self.Xf if D is None else np.fft.rfft(np.asarray(D), axis=0)
            Sf = np.sum(self.Zf * Df, axis=self.cri.axisM - 1)
            return np.fft.irfft(Sf, self.cri.Nv[1], axis=0).astype(self.dtype)
        Df = self.Xf if D is None else np.fft.rfftn(np.asarray(D), axes=(0, 1))
        Sf = np.sum(self.Zf * Df, axis=self.cri.axisM)
        return np.fft.irfftn(Sf, self.cri.Nv, axes=(0, 1)).astype(self.dtype)


class ConvCnstrMODMask(ConvCnstrMOD):
    r"""PGM dictionary update with a spatial mask in the data fidelity term,
    (1/2) sum_k ||W (sum_m d_m * x_{k,m} - s_k)||_2^2 over constrained filters (reference
    class: sporco/pgm/ccmod.py:408-604).  ``W`` must be compatible with the *internal* layout
    of ``S``, (H, W, C, K, 1) (single-channel dictionaries: (H, W, 1, C K, 1) works too)."""

    _volumes_ok = False

    def __init__(self, Z, S, W, dsz, opt=None, dimK=None, dimN=2, **backend):
        if opt is None:
            opt = ConvCnstrMODMask.Options()
        W = np.asarray(W)
        if dimN == 1:
            # (signals: a unit first axis on every array, see ConvCnstrMOD)
            S, dsz, dimN = np.asarray(S)[np.newaxis], _dsz_unit_axis(dsz), 2
            W = W[np.newaxis] if W.ndim > 0 else W
            if Z is not None and np.ndim(Z) == 4:
                Z = np.asarray(Z)[np.newaxis]
            backend['_dim1'] = True
        cri = cr.CDU_ConvRepIndexing(dsz, S, dimK=dimK, dimN=dimN)
        if W.ndim < dimN + 3:
            W = W.reshape(W.shape + (1,) * (dimN + 3 - W.ndim))
        if cri.C > 1 and cri.Cd == 1:
            # channels fold into the image axis, as S does (pgm/ccmod.py:514-528)
            shp = list(W.shape)
            if 1 < shp[cri.axisC] * shp[cri.axisK] < cri.C * cri.K:
                if shp[cri.axisK] == 1 and cri.K > 1:
                    shp[cri.axisK] = cri.K
                else:
                    shp[cri.axisC] = cri.C
                W = np.broadcast_to(W, shp)
            W = W.reshape(W.shape[0:dimN] + (1, W.shape[cri.axisC] * W.shape[cri.axisK], 1))
        self.W = W
        super(ConvCnstrMODMask, self).__init__(Z, S, dsz, opt, dimK=dimK, dimN=dimN, **backend)
        self.W = np.asarray(self.W, dtype=self.dtype)
        H, Wd = self.cri.Nv
        w = self.W.reshape(self.W.shape[0:2] + (1,) * (5 - self.W.ndim) + self.W.shape[2:]) \
            if self.W.ndim < 5 else self.W
        # device layout of the folded image axis: (C, K) row-major == (1, C K)
        full = (H, Wd, self.cri.C, self.cri.K, 1)
        if w.shape[2] == 1 and w.shape[3] == self.cri.C * self.cri.K and self.cri.C > 1:
            w = w.reshape(w.shape[0], w.shape[1], self.cri.C, self.cri.K, 1)
        for ws, fs in zip(w.shape, full):
            if ws not in (1, fs):
                raise ValueError("mask of shape %s cannot broadcast to %s" % (w.shape, full))
        self.dev.set_data_mask(np.ascontiguousarray(w))

    def grad_f(self, V=None):
        """sum_k conj(Zf) rfftn(W^2 irfftn(sum_m Zf V - Sf)) (pgm/ccmod.py:552-575)."""
        if V is None:
            V = _lib.VAR_DYF
        self.dev.masked_grad(V, True, True)
        if self._reducer is not None:
            self._reducer.all_reduce_array(self.dev, _lib.VAR_DGF)
        self._fcache.pop(V, None)
        self.invalidate(_lib.VAR_DGF)
        return _lib.VAR_DGF

    def _masked_eval(self, var):
        out = self.dev.masked_grad(var, True, False)
        return out if self._reducer is None else self._reducer.sum(out)

    def obfn_dfd(self):
        """(1/2) ||W irfftn(sum_m Zf Xf - Sf)||^2 (pgm/ccmod.py:579-587)."""
        return self._masked_eval(_lib.VAR_DXF)[_lib.PGM_DFID] / 2.0

    def obfn_f(self, Xf=None):
        """(1/2) ||rfftn(W irfftn(sum_m Zf Xf - Sf))||^2 (pgm/ccmod.py:591-604)."""
        return self._masked_eval(_lib.VAR_DXF if Xf is None else Xf)[_lib.PGM_F]
