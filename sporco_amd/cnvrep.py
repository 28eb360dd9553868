"""Shape inference and dictionary constraint helpers (host side).

Same inference rules as ``sporco.cnvrep`` (sporco/cnvrep.py:24-198 for sparse
coding, :277-454 for dictionary update, :492-550 for weight shapes,
:609-1074 for the constraint projection), restated compactly.  Internal
layout: ``S(N0, N1, C, K, 1)``, ``D(N0, N1, Cd, 1, M)``, ``X(N0, N1, Cx, K, M)``.
"""

import pprint

import numpy as np


class CSC_ConvRepIndexing(object):
    """Problem dimensions of a convolutional sparse coding problem."""

    def __init__(self, D, S, dimK=None, dimN=2):
        self.dimCd = D.ndim - (dimN + 1)
        self.Cd = 1 if self.dimCd == 0 else D.shape[-2]
        extra = S.ndim - dimN
        if dimK is None:
            if extra == 0:
                dimC, dimK = 0, 0
            elif extra == 1:
                # a single extra axis follows the dictionary: channels if D has them
                dimC = self.dimCd
                dimK = 1 - dimC
            else:
                dimC, dimK = 1, 1
        else:
            dimC = extra - dimK
        self.dimN, self.dimC, self.dimK = dimN, dimC, dimK
        self.C = S.shape[dimN] if dimC == 1 else 1
        if self.Cd > 1 and self.C != self.Cd:
            raise ValueError("Multi-channel dictionary with signal with mismatched "
                             "number of channels (Cd=%d, C=%d)" % (self.Cd, self.C))
        self.K = S.shape[dimN + dimC] if dimK == 1 else 1
        self.M = D.shape[-1]
        self.Nv = S.shape[0:dimN]
        self.N = int(np.prod(np.array(self.Nv)))
        self.axisN = tuple(range(dimN))
        self.axisC, self.axisK, self.axisM = dimN, dimN + 1, dimN + 2
        Cx = self.C - self.Cd + 1
        self.shpD = D.shape[0:dimN] + (self.Cd, 1, self.M)
        self.shpS = self.Nv + (self.C, self.K, 1)
        self.shpX = self.Nv + (Cx, self.K, self.M)

    def __str__(self):
        return pprint.pformat(vars(self))


class DictionarySize(object):
    """Parameters of a dictionary size tuple ``dsz`` (sporco/cnvrep.py:211-264): one scale
    ``(rows, cols, [channels,] filters)``, or a tuple of such blocks for a multi-scale
    dictionary.  ``fsz``: the (rows, cols) support of every filter for a multi-scale ``dsz``
    (what the device's constraint projection is told), else None.  The nested form with
    separate channel blocks (cnvrep.py:773-782) is not taken."""

    def __init__(self, dsz, dimN=2):
        self.dsz = dsz
        self.fsz = None
        if isinstance(dsz[0], tuple):
            if isinstance(dsz[0][0], tuple):
                raise NotImplementedError("dictionary size blocks with separate channel blocks "
                                          "are outside the sporco_amd hot path")
            self.ndim = len(dsz[0])
            self.nchn = 1 if self.ndim == dimN + 1 else dsz[0][-2]
            mxsz = np.zeros((dimN,), dtype=int)
            self.nflt = 0
            self.fsz = []
            for blk in dsz:
                if len(blk) != self.ndim or (self.ndim > dimN + 1 and blk[-2] != self.nchn):
                    raise ValueError("dictionary size blocks of different kinds")
                mxsz = np.maximum(mxsz, np.asarray(blk[0:dimN]))
                self.nflt += blk[-1]
                self.fsz += [tuple(int(v) for v in blk[0:dimN])] * int(blk[-1])
            self.mxsz = tuple(int(v) for v in mxsz)
        else:
            self.ndim = len(dsz)
            self.mxsz = tuple(dsz[0:dimN])
            self.nflt = dsz[-1]
            self.nchn = 1 if self.ndim == dimN + 1 else dsz[-2]

    def __str__(self):
        return pprint.pformat(vars(self))


class CDU_ConvRepIndexing(object):
    """Problem dimensions of a convolutional dictionary update problem
    (sporco/cnvrep.py:277-454)."""

    def __init__(self, dsz, S, dimK=None, dimN=2):
        ds = DictionarySize(dsz, dimN)
        self.dimCd = ds.ndim - dimN - 1
        self.Cd = ds.nchn
        self.M = ds.nflt
        self.dsz = dsz
        self.mxsz = ds.mxsz     # largest filter support (= dsz[0:dimN] for one scale)
        self.fsz = ds.fsz       # per-filter supports of a multi-scale dsz, else None
        if dimK is None:
            rdim = S.ndim - dimN
            if rdim == 0:
                dimC, dimK = 0, 0
            elif rdim == 1:
                dimC = self.dimCd
                dimK = S.ndim - dimN - dimC
            else:
                dimC, dimK = 1, 1
        else:
            dimC = S.ndim - dimN - dimK
        self.dimN, self.dimC, self.dimK = dimN, dimC, dimK
        self.C = S.shape[dimN] if dimC == 1 else 1
        self.Cx = self.C - self.Cd + 1
        if self.Cd > 1 and self.C != self.Cd:
            raise ValueError("Multi-channel dictionary with signal with mismatched "
                             "number of channels (Cd=%d, C=%d)" % (self.Cd, self.C))
        self.K = S.shape[dimN + dimC] if dimK == 1 else 1
        self.Nv = S.shape[0:dimN]
        self.N = int(np.prod(np.array(self.Nv)))
        self.axisN = tuple(range(dimN))
        self.axisC, self.axisK, self.axisM = dimN, dimN + 1, dimN + 2
        self.shpD = self.Nv + (self.Cd, 1, self.M)
        self.shpS = self.Nv + (self.C, self.K, 1)
        self.shpX = self.Nv + (self.Cx, self.K, self.M)

    def __str__(self):
        return pprint.pformat(vars(self))


def stdformD(D, Cd, M, dimN=2):
    """Reshape a dictionary to the internal (.., Cd, 1, M) layout."""
    return D.reshape(D.shape[0:dimN] + (Cd, 1, M))


def l1Wshape(W, cri):
    """Internal shape of an ``L1Weight`` array (sporco/cnvrep.py:492-550)."""
    sdim = cri.dimN + cri.dimC + cri.dimK
    if W.ndim < sdim:
        if W.size != 1:
            raise ValueError('weight array must be scalar or have at least '
                             'the same number of dimensions as input array')
        return (1,) * (cri.dimN + 3)
    if W.ndim == sdim:
        return W.shape + (1,) * (3 - cri.dimC - cri.dimK)
    if W.ndim == cri.dimN + 3:
        return W.shape
    # otherwise the last axis is taken to be the filter index
    return W.shape[0:-1] + (1,) * (2 - cri.dimC - cri.dimK) + W.shape[-1:]


def mskWshape(W, cri):
    """Internal shape of a spatial mask array (sporco/cnvrep.py:554-605)."""
    ckdim = W.ndim - cri.dimN
    if ckdim >= 2:
        shp = W.shape + (1,) if ckdim == 2 else W.shape
    elif ckdim == 1:
        if cri.C == 1 and cri.K > 1:
            shp = W.shape[0:cri.dimN] + (1, W.shape[cri.dimN], 1)
        elif cri.C > 1 and cri.K == 1:
            shp = W.shape[0:cri.dimN] + (W.shape[cri.dimN], 1, 1)
        else:
            shp = W.shape[0:cri.dimN] + (W.shape[cri.dimN], 1, 1)
    else:
        shp = W.shape + (1,) * 3
    return shp


# -- dictionary constraint set -------------------------------------------------

def fold3(a, Dz, Hs):
    """(depth, height, ...) -> (depth * height, ...): how the dimN = 3 arrays reach the device (the
    memory layout is the same); arrays that broadcast along both axes lose one of the unit axes."""
    if a.shape[0:2] == (Dz, Hs):
        return a.reshape((Dz * Hs,) + a.shape[2:])
    if a.shape[0:2] == (1, 1):
        return a[0]
    raise ValueError("dimN = 3: an array of shape %s neither spans nor broadcasts along the first "
                     "two axes" % (a.shape,))


def unfold3(a, Dz, Hs):
    return a.reshape((Dz, Hs) + a.shape[1:]) if a.ndim >= 1 and a.shape[0] == Dz * Hs else a


def volume_problem(D, S, dimK):
    """The folded two-dimensional problem of a dimN = 3 one: ((depth, height), folded zero-padded
    single-channel dictionary (depth * height, W, M), folded signal (depth * height, W, C, K))."""
    D, S = np.asarray(D), np.asarray(S)
    c3 = CSC_ConvRepIndexing(D, S, dimK=dimK, dimN=3)
    if c3.Cd > 1:
        raise NotImplementedError("dimN = 3: single-channel dictionary")
    Dz, Hs = int(c3.Nv[0]), int(c3.Nv[1])
    D2 = fold3(zpad(D.reshape(c3.shpD), c3.Nv), Dz, Hs)[:, :, 0, 0]
    S2 = fold3(S.reshape(c3.shpS), Dz, Hs)[..., 0]
    return (Dz, Hs), D2, S2


def zpad(v, Nv):
    """Zero-pad the leading axes of ``v`` to ``Nv``."""
    out = np.zeros(tuple(Nv) + v.shape[len(Nv):], dtype=v.dtype)
    out[tuple(slice(0, n) for n in v.shape)] = v
    return out


def _size_blocks(dsz, dimN):
    """(filter slice, support) of every block of a multi-scale ``dsz``."""
    m0 = 0
    for blk in dsz:
        if isinstance(blk[0], tuple):
            raise NotImplementedError("dictionary size blocks with separate channel blocks are "
                                      "outside the sporco_amd hot path")
        m1 = m0 + blk[-1]
        yield slice(m0, m1), tuple(slice(0, n) for n in blk[0:-1])
        m0 = m1


def bcrop(v, dsz, dimN=2):
    """Crop to the filter support; a multi-scale ``dsz`` gives an array of the largest support
    with every filter zero outside its own (sporco/cnvrep.py:729-817)."""
    if isinstance(dsz[0], tuple):
        mx = DictionarySize(dsz, dimN).mxsz
        vc = np.zeros(tuple(mx) + v.shape[dimN:], dtype=v.dtype)
        for fsl, sup in _size_blocks(dsz, dimN):
            idx = sup + (Ellipsis, fsl)
            vc[idx] = v[idx]
        return vc
    return v[tuple(slice(0, n) for n in dsz[0:dimN])]


def zeromean(v, dsz, dimN=2):
    """Subtract, per filter (and channel), the mean over the filter's support
    (sporco/cnvrep.py:609-670)."""
    out = v.copy()
    axisN = tuple(range(dimN))
    if isinstance(dsz[0], tuple):
        for fsl, sup in _size_blocks(dsz, dimN):
            idx = sup + (Ellipsis, fsl)
            out[idx] -= np.mean(v[idx], axisN)
        return out
    sup = tuple(slice(0, n) for n in dsz[0:dimN])
    out[sup] -= np.mean(v[sup], axisN)
    return out


def normalise(v, dimN=2):
    """Unit l2 norm over the first ``dimN`` axes (zero vectors untouched)."""
    nrm = np.sqrt(np.sum(np.abs(v) ** 2, tuple(range(dimN)), keepdims=True))
    nrm[nrm == 0] = 1.0
    return np.asarray(v / nrm, dtype=v.dtype)


def Pcn(x, dsz, Nv, dimN=2, dimC=1, crp=False, zm=False):
    """Projection onto the constraint set: crop, (zero-pad), (zero-mean), normalise
    (sporco/cnvrep.py:868-913)."""
    v = bcrop(x, dsz, dimN)
    if not crp:
        v = zpad(v, Nv)
    if zm:
        v = zeromean(v, dsz, dimN)
    return normalise(v, dimN + dimC)


def getPcn(dsz, Nv, dimN=2, dimC=1, crp=False, zm=False):
    """Return ``x -> Pcn(x, ...)`` with the options bound."""
    def proj(x):
        return Pcn(x, dsz, Nv, dimN, dimC, crp, zm)
    return proj
