"""CPU oracle for the ConvBPDN hot path -- TEST INFRASTRUCTURE ONLY.

This module is a plain-NumPy restatement of the arithmetic performed by the
reference (bwohlberg/sporco, mounted read-only at /root/reference) on the
FFT-domain convolutional sparse coding path.  It exists so that the HIP
kernels of ``sporco_amd`` can be checked on a GPU box where the reference
itself is not present.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product package never
does (``sporco_amd`` fails loudly when its HIP library is missing).

Pinning: ``oracle/make_golden.py`` runs the *unmodified* reference in the
authoring container and stores its outputs under ``tests/golden/``;
``tests/test_oracle_vs_golden.py`` checks every function here against those
fixtures (so parity is pinned to the reference itself, to tolerance --
the reference path is floating point and its own tests pin it to tolerance
only).

All arrays use the reference's internal 5-D layout
``(H, W, C, N, K)`` = SPORCO ``(N0, N1, C, K, M)`` (``sporco/cnvrep.py:86-111``),
filter index fastest.  Letters follow BASELINE.json: N images, K filters.

Every function cites the reference lines it follows.
"""

import time

import numpy as np

AX_SPATIAL = (0, 1)   # cri.axisN for dimN = 2   (sporco/cnvrep.py:190)
AX_C = 2              # cri.axisC                 (sporco/cnvrep.py:191)
AX_N = 3              # cri.axisK (image index)   (sporco/cnvrep.py:192)
AX_K = 4              # cri.axisM (filter index)  (sporco/cnvrep.py:193)


# ---------------------------------------------------------------------------
# dtype helpers (sporco/fft.py:44-101 complex_dtype / real_dtype)
# ---------------------------------------------------------------------------

def complex_dtype(dtype):
    return (np.zeros(1, dtype=dtype) + 1j).dtype


def real_dtype(dtype):
    return np.zeros(1, dtype=dtype).real.dtype


# ---------------------------------------------------------------------------
# FFT primitives (sporco/fft.py:257-314; numpy fallback :631-639)
# ---------------------------------------------------------------------------

# Threads of the FFTs.  None: numpy.fft, the reference's fallback and what every parity test
# runs.  An int: scipy.fft (pocketfft) with that many workers -- the counterpart of the
# reference's normal install, which runs its FFTs in pyFFTW on multiprocessing.cpu_count()
# threads (sporco/fft.py:37, call sites :222-312) while the rest stays single-threaded NumPy;
# used only by bench.py's threaded cpu_baseline leg.
FFT_WORKERS = None


def rfftn2(a, s=None):
    """Unnormalised 2-D real FFT over axes (0, 1), result cast to the complex
    type matching ``a`` -- sporco/fft.py:631-634 (`_rfftn`)."""
    if FFT_WORKERS:
        import scipy.fft as sfft
        return sfft.rfftn(a, s, AX_SPATIAL, workers=FFT_WORKERS).astype(complex_dtype(a.dtype),
                                                                       copy=False)
    return np.fft.rfftn(a, s, AX_SPATIAL).astype(complex_dtype(a.dtype))


def irfftn2(af, s):
    """Inverse of :func:`rfftn2`; ``s`` = (H, W) is mandatory because W may be
    odd -- sporco/fft.py:636-639 (`_irfftn`)."""
    if FFT_WORKERS:
        import scipy.fft as sfft
        return sfft.irfftn(af, s, AX_SPATIAL, workers=FFT_WORKERS).astype(real_dtype(af.dtype),
                                                                         copy=False)
    return np.fft.irfftn(af, s, AX_SPATIAL).astype(real_dtype(af.dtype))


def rfl2norm2(xf, xs):
    """Squared l2 norm of the spatial-domain array whose rfftn2 is ``xf``
    (half-spectrum Parseval with weights 1,2,...,2,(1)) -- sporco/fft.py:449-484.
    ``xs`` is the spatial shape tuple (only xs[0], xs[1] are used)."""
    scl = 1.0 / (xs[0] * xs[1])
    w = xs[1]
    n0 = np.linalg.norm(xf[:, 0])
    idx1 = (w + 1) // 2
    n1 = np.linalg.norm(xf[:, 1:idx1])
    n2 = np.linalg.norm(xf[:, -1:]) if w % 2 == 0 else 0.0
    return scl * (n0 ** 2 + 2.0 * n1 ** 2 + n2 ** 2)


# ---------------------------------------------------------------------------
# linalg primitives (sporco/linalg.py)
# ---------------------------------------------------------------------------

def inner(x, y, axis=-1):
    """sum(x*y) over ``axis`` with keepdims, no conjugation --
    sporco/linalg.py:41-88."""
    return np.sum(x * y, axis=axis, keepdims=True)


def solvedbi_sm_c(ah, a, rho, axis=AX_K):
    """c = ah / (ah.a + rho) -- sporco/linalg.py:277-297."""
    return ah / (inner(ah, a, axis=axis) + rho)


def solvedbi_sm(ah, rho, b, c=None, axis=AX_K):
    """Solve (rho I + a a^H) x = b per pixel by Sherman-Morrison --
    sporco/linalg.py:232-273."""
    a = np.conj(ah)
    if c is None:
        c = solvedbi_sm_c(ah, a, rho, axis)
    return (b - (a * inner(c, b, axis=axis))) / rho


def solvedbd_sm_c(ah, a, d, axis=AX_K):
    """(ah/d) / (<ah, a/d> + 1) -- sporco/linalg.py:346-366."""
    return (ah / d) / (inner(ah, a / d, axis=axis) + 1.0)


def solvedbd_sm(ah, d, b, c=None, axis=AX_K):
    """Solve (diag(d) + a a^H) x = b per frequency -- sporco/linalg.py:300-342."""
    a = np.conj(ah)
    if c is None:
        c = solvedbd_sm_c(ah, a, d, axis)
    return (b - a * inner(c, b, axis=axis)) / d


def solvemdbi_ism(ah, rho, b, axisM=AX_K, axisK=AX_C):
    """Solve (rho I + sum_c a_c a_c^H) x = b per frequency by iterated
    Sherman-Morrison over the C rank-one terms -- sporco/linalg.py:370-444
    (``axisK`` is the reference's name for the axis indexing the terms: here the
    dictionary channel axis)."""
    C = ah.shape[axisK]
    a = np.conj(ah)
    take = lambda v, c: np.take(v, [c], axisK)
    gamma, delta = [], []
    alpha = take(a, 0) / rho
    beta = b / rho
    for c in range(C):
        gamma.append(alpha.copy())
        delta.append(1.0 + inner(take(ah, c), gamma[c], axis=axisM))
        beta = beta - gamma[c] * inner(take(ah, c), beta, axis=axisM) / delta[c]
        if c < C - 1:
            alpha = take(a, c + 1) / rho
            for l in range(c + 1):
                alpha = alpha - gamma[l] * inner(take(ah, l), alpha, axis=axisM) / delta[l]
    return beta


def gradient_filters_ghg(shp, dtype):
    """sum_i |G_i|^2 on the half spectrum, shape (H, W//2+1, 1, 1, 1), for the
    two-tap difference filters [1, -1] along each spatial axis --
    sporco/signal.py:204-240 (the Gf themselves are only used by the
    reference through this sum, sporco/admm/cbpdn.py:1141-1143)."""
    H, W = shp
    g = np.zeros((2, 2, 1, 1, 1, 2), dtype=dtype)
    g[:, 0, 0, 0, 0, 0] = (1, -1)
    g[0, :, 0, 0, 0, 1] = (1, -1)
    Gf = np.fft.rfftn(g, (H, W), axes=(0, 1))
    return np.sum(np.conj(Gf) * Gf, axis=-1).real


def rrs(ax, b):
    """Relative residual ||b - ax|| / max(||ax||, ||b||) --
    sporco/linalg.py:883-910."""
    nrm = max(np.linalg.norm(ax.ravel()), np.linalg.norm(b.ravel()))
    if nrm == 0.0:
        return 0.0
    return np.linalg.norm((ax - b).ravel()) / nrm


# ---------------------------------------------------------------------------
# proximal operators (sporco/prox/_lp.py, sporco/prox/_l21.py, sporco/array.py)
# ---------------------------------------------------------------------------

def prox_l1(v, alpha):
    """Soft threshold sign(v) * max(|v| - alpha, 0), real input --
    sporco/prox/_lp.py:174-181."""
    return np.sign(v) * np.clip(np.abs(v) - alpha, 0, float('Inf'))


def zdivide(x, y):
    """x / y with 0 where y == 0 -- sporco/array.py:119-137."""
    return np.divide(x, y, out=np.zeros_like(x), where=(y != 0))


def prox_l2(v, alpha, axis=None):
    """Vector shrinkage v/||v|| * max(0, ||v|| - alpha) over ``axis`` --
    sporco/prox/_lp.py:283-290."""
    a = np.sqrt(np.sum(v ** 2, axis=axis, keepdims=True))
    b = np.maximum(0, a - alpha)
    b = zdivide(b, a)
    return np.asarray(b * v, dtype=v.dtype)


def prox_sl1l2(v, alpha, beta, axis=None):
    """prox of alpha*l1 + beta*l2 = S_2,beta(S_1,alpha(v)) --
    sporco/prox/_l21.py:51-88."""
    return prox_l2(prox_l1(v, alpha), beta, axis)


# ---------------------------------------------------------------------------
# ADMM ConvBPDN / ConvBPDNJoint (sporco/admm/cbpdn.py + sporco/admm/admm.py)
# ---------------------------------------------------------------------------

def default_rho_xi(lmbda):
    """rho_xi default -- sporco/admm/cbpdn.py:588-591."""
    if lmbda != 0.0:
        return float(1.0 + 18.3 ** (np.log10(lmbda) + 1.0))
    return 1.0


def admm_cbpdn(D, S, lmbda, mu=None, dtype=np.float32, maxiter=50,
               rho=None, rlx=1.8, auto_rho=True, rho_period=1,
               rho_tau=1000.0, rho_mu=1.2, rho_xi=None, auto_scaling=True,
               std_residuals=False, abs_tol=0.0, rel_tol=1e-3,
               nonneg=False, nobndry=False, wl1=1.0, wl21=1.0,
               gevaly=False, fevalx=True, stats=True, Y0=None, U0=None,
               time_budget=None, grad_mu=None, grad_weight=1.0, ams_mask=None):
    """Run ADMM ConvBPDN (``mu is None``) or ConvBPDNJoint on 5-D arrays.

    ``D``: (dH, dW, Cd, 1, K);  ``S``: (H, W, C, N, 1).  Cd = 1 is the
    ``solvedbi_sm`` branch of sporco/admm/cbpdn.py:274-276; Cd = C > 1 (a
    multi-channel dictionary, one coefficient map set shared by the channels:
    X is (H, W, 1, N, K)) the ``solvemdbi_ism`` branch (:250-251, :277-279).

    Follows, per iteration, sporco/admm/admm.py:331-377:
      Yprev=Y; xstep (cbpdn.py:267-281); relax_AX (admm.py:877-885);
      ystep (cbpdn.py:614-620 / :785-794 then :297-311); ustep (admm.py:434-437);
      compute_residuals (admm.py:462-486 with :959-983);
      eval_objfn (cbpdn.py:325-344, :624-630, :798-807);
      update_rho (admm.py:549-575); stop test (admm.py:375-377).

    ``grad_mu`` (with ``mu is None``) selects ConvBPDNGradReg
    (cbpdn.py:992-1214): the x step solves with diagonal
    ``grad_mu * grad_weight[k] * GHGf + rho`` (:1167-1175) and the objective
    gains ``grad_mu * RegGrad`` (:1204-1214).

    ``ams_mask`` selects the additive-mask-simulation wrapper AddMaskSim
    (cbpdn.py:2287-2485) around the chosen class: ``D`` must already carry the
    appended impulse filter as its last filter (:2345-2353; one per channel as the
    last Cd filters for a multi-channel dictionary), ``ams_mask`` is the
    mask W in its internal 5-D shape (cnvrep.mskWshape, channels swapped onto the
    filter axis when Cd > 1, :2358-2364); the y step leaves the
    impulse slice unshrunk and zeroes it where W is nonzero (:2378-2394) and the
    regularisers ignore that slice (:2398-2412).

    Returns a dict with final X, Y, U, Xf and the per-iteration traces.
    """
    dtype = np.dtype(dtype)
    rdt = real_dtype(dtype).type
    D = np.asarray(D, dtype=dtype)
    S = np.asarray(S, dtype=dtype)
    H, W = S.shape[0], S.shape[1]
    K = D.shape[AX_K]
    mcd = D.shape[AX_C] > 1
    shpX = (H, W, 1 if mcd else S.shape[AX_C], S.shape[AX_N], K)
    Nx = int(np.prod(shpX))

    lmbda = rdt(lmbda)
    joint = mu is not None
    if joint:
        mu = dtype.type(mu)
    # penalty parameter default: cbpdn.py:584
    rho = rdt(50.0 * lmbda + 1.0) if rho is None else rdt(rho)
    if rho_xi is None:
        rho_xi = default_rho_xi(lmbda)
    rho_xi = rdt(rho_xi)
    rho_tau = rdt(rho_tau)
    rho_mu = rdt(rho_mu)
    rlx = rdt(rlx)
    wl1 = np.asarray(wl1, dtype=real_dtype(dtype))
    wl21 = np.asarray(wl21, dtype=dtype)

    # cbpdn.py:231, :247-249
    Sf = rfftn2(S)
    Df = rfftn2(D, (H, W))
    DSf = np.conj(Df) * Sf
    if mcd:
        DSf = np.sum(DSf, axis=AX_C, keepdims=True)        # cbpdn.py:250-251
    # AddMaskSim: one impulse filter, or one per channel of a multi-channel dictionary
    # (cbpdn.py:2337-2346; index_addmsk :2447-2452)
    n_imp = D.shape[AX_C] if mcd else 1
    gradreg = grad_mu is not None
    if gradreg:
        assert not joint
        grad_mu = dtype.type(grad_mu)                      # cbpdn.py:1133
        wg = np.asarray(grad_weight, dtype=dtype)
        if wg.ndim:
            wg = wg.reshape((1, 1, 1, 1) + wg.shape)       # cbpdn.py:1134-1139
        GHGf = wg * gradient_filters_ghg((H, W), dtype)    # cbpdn.py:1141-1143

    Y = np.zeros(shpX, dtype=dtype) if Y0 is None else \
        np.asarray(Y0).astype(dtype, copy=True)
    if U0 is not None:
        U = np.asarray(U0).astype(dtype, copy=True)
    elif Y0 is None:
        U = np.zeros(shpX, dtype=dtype)
    else:
        U = (lmbda / rho) * np.sign(Y)           # cbpdn.py:601-610

    tr = {k: [] for k in ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'RegGrad', 'PrimalRsdl',
                          'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')}
    X = None
    Xf = None
    t0 = time.perf_counter()
    k = 0
    for k in range(maxiter):
        Yprev = Y.copy()
        # -- xstep: cbpdn.py:267-281
        YU = Y - U
        b = DSf + rho * rfftn2(YU)
        if gradreg and mcd:
            # cbpdn.py:1181-1184: the iterated solve with the diagonal in place of rho
            Xf = solvemdbi_ism(Df, grad_mu * GHGf + rho, b, AX_K, AX_C).astype(b.dtype)
        elif gradreg:
            Xf = solvedbd_sm(Df, grad_mu * GHGf + rho, b, None,
                             AX_K).astype(b.dtype)
        elif mcd:
            Xf = solvemdbi_ism(Df, rho, b, AX_K, AX_C).astype(b.dtype)
        else:
            Xf = solvedbi_sm(Df, rho, b, None, AX_K).astype(b.dtype)
        X = irfftn2(Xf, (H, W))
        # -- relax_AX: admm.py:877-885
        AXnr = X
        AX = X if rlx == 1.0 else rlx * X + (1 - rlx) * Y
        # -- ystep
        if ams_mask is not None:
            Yi = (AX + U)[..., -n_imp:].copy()           # cbpdn.py:2386-2387
        if joint:
            Y = prox_sl1l2(AX + U, (lmbda / rho) * wl1, (mu / rho) * wl21,
                           axis=AX_C)
        else:
            Y = prox_l1(AX + U, (lmbda / rho) * wl1)
        if nonneg:
            Y[Y < 0.0] = 0.0
        if nobndry:
            Y[1 - D.shape[0]:] = 0.0
            Y[:, 1 - D.shape[1]:] = 0.0
        if ams_mask is not None:
            # (index semantics of the reference kept as they are: np.where on W's own
            # shape, so a broadcast axis of W addresses index 0 only)
            Yi[np.where(np.asarray(ams_mask).astype(bool))] = 0.0   # cbpdn.py:2393
            Y[..., -n_imp:] = Yi
        # -- ustep: admm.py:434-437
        U = U + (AX - Y)
        if stats or auto_rho:
            # -- compute_residuals: admm.py:462-486
            nAX, nY = np.linalg.norm(AXnr), np.linalg.norm(Y)
            nr = np.linalg.norm(AXnr - Y)
            ns = np.linalg.norm(rho * (Yprev - Y))
            if std_residuals:
                r, s = nr, ns
                epri = np.sqrt(Nx) * abs_tol + max(nAX, nY) * rel_tol
                edua = np.sqrt(Nx) * abs_tol + rho * np.linalg.norm(U) * rel_tol
            else:
                rn = max(nAX, nY)
                rn = 1.0 if rn == 0.0 else rn
                sn = rho * np.linalg.norm(U)
                sn = 1.0 if sn == 0.0 else sn
                r, s = nr / rn, ns / sn
                epri = np.sqrt(Nx) * abs_tol / rn + rel_tol
                edua = np.sqrt(Nx) * abs_tol / sn + rel_tol
        if stats:
            # -- eval_objfn: cbpdn.py:325-344
            fvar = Xf if fevalx else rfftn2(Y)
            Ef = inner(Df, fvar, axis=AX_K) - Sf
            dfd = rfl2norm2(Ef, S.shape) / 2.0
            gvar = Y if gevaly else X
            if ams_mask is not None:
                gvar = gvar.copy()
                gvar[..., -n_imp:] = 0                   # cbpdn.py:2404-2411
            rl1 = np.linalg.norm((wl1 * gvar).ravel(), 1)
            if joint:
                rl21 = np.sum(wl21 * np.sqrt(np.sum(gvar ** 2, axis=AX_C)))
                obj = dfd + lmbda * rl1 + mu * rl21
            elif gradreg:
                rl21 = 0.0
                rgr = rfl2norm2(np.sqrt(GHGf * np.conj(fvar) * fvar),
                                S.shape) / 2.0
                obj = dfd + lmbda * rl1 + grad_mu * rgr
            else:
                rl21 = 0.0
                obj = dfd + lmbda * rl1
            for key, val in (('ObjFun', obj), ('DFid', dfd), ('RegL1', rl1),
                             ('RegL21', rl21),
                             ('RegGrad', rgr if gradreg else 0.0),
                             ('PrimalRsdl', r),
                             ('DualRsdl', s), ('EpsPrimal', epri),
                             ('EpsDual', edua), ('Rho', rho)):
                tr[key].append(float(val))
        # -- update_rho: admm.py:549-575
        if auto_rho and k != 0 and (k + 1) % rho_period == 0:
            if auto_scaling:
                if s == 0.0 or r == 0.0:
                    rhomlt = rho_tau
                else:
                    rhomlt = np.sqrt(r / (s * rho_xi) if r > s * rho_xi
                                     else (s * rho_xi) / r)
                    if rhomlt > rho_tau:
                        rhomlt = rho_tau
            else:
                rhomlt = rho_tau
            rsf = 1.0
            if r > rho_xi * rho_mu * s:
                rsf = rhomlt
            elif s > (rho_mu / rho_xi) * r:
                rsf = 1.0 / rhomlt
            rho = rho * rdt(rsf)
            U = U / rsf
        if (stats or auto_rho) and r < epri and s < edua:
            break
        if time_budget is not None and time.perf_counter() - t0 > time_budget:
            break
    elapsed = time.perf_counter() - t0
    out = {key: np.array(val) for key, val in tr.items()}
    out.update(X=X, Y=Y, U=U, Xf=Xf, Df=Df, Sf=Sf, rho=rho, iters=k + 1,
               seconds=elapsed)
    return out


def reconstruct(Df, X, s):
    """irfftn(sum_k Df * rfftn(X)) -- sporco/admm/cbpdn.py:373-380."""
    Xf = rfftn2(X)
    return irfftn2(np.sum(Df * Xf, axis=AX_K), s)


# ---------------------------------------------------------------------------
# PGM (FISTA) ConvBPDN (sporco/pgm/cbpdn.py + sporco/pgm/pgm.py)
# ---------------------------------------------------------------------------

def pgm_cbpdn(D, S, lmbda, dtype=np.float32, maxiter=50, L=500.0,
              rel_tol=1e-3, nonneg=False, nobndry=False, wl1=1.0,
              stats=True, time_budget=None):
    """FISTA ConvBPDN with Nesterov momentum, fixed L (no backtracking).

    Follows sporco/pgm/pgm.py:328-370 with PGMDFT.xstep (:779-811),
    PGMDFT.ystep (:815-831), MomentumNesterov.update
    (sporco/pgm/momentum.py:65-70), grad_f/prox_g/rsdl/eval_objfn of
    sporco/pgm/cbpdn.py:263-356.
    """
    dtype = np.dtype(dtype)
    D = np.asarray(D, dtype=dtype)
    S = np.asarray(S, dtype=dtype)
    H, W = S.shape[0], S.shape[1]
    K = D.shape[AX_K]
    mcd = D.shape[AX_C] > 1        # multi-channel dictionary: one coefficient channel
    shpX = (H, W, 1 if mcd else S.shape[AX_C], S.shape[AX_N], K)
    lmbda = dtype.type(lmbda)
    L = dtype.type(L)
    wl1 = np.asarray(wl1, dtype=dtype)

    Sf = rfftn2(S)
    Df = rfftn2(D, (H, W))
    X = np.zeros(shpX, dtype=dtype)
    Xf = rfftn2(X)
    Yf = Xf.copy()
    Yfprv = Yf.copy() + 1e5                        # pgm/cbpdn.py:233
    t = 1.0
    tr = {k: [] for k in ('ObjFun', 'DFid', 'RegL1', 'Rsdl')}
    t0 = time.perf_counter()
    k = 0
    for k in range(maxiter):
        Xfprv = Xf.copy()                           # pgm.py:835-846
        if stats:
            Yfprv = Yf.copy()
        # xstep: pgm.py:779-811
        gradf = np.conj(Df) * (inner(Df, Yf, axis=AX_K) - Sf)
        if mcd:
            gradf = np.sum(gradf, axis=AX_C, keepdims=True)   # pgm/cbpdn.py:277-279
        Vf = Yf - (1.0 / L) * gradf
        V = irfftn2(Vf.astype(complex_dtype(dtype)), (H, W))
        X = prox_l1(V, (lmbda / L) * wl1)
        if nonneg:
            X[X < 0.0] = 0.0
        if nobndry:
            X[1 - D.shape[0]:] = 0.0
            X[:, 1 - D.shape[1]:] = 0.0
        Xf = rfftn2(X)
        # ystep: pgm.py:815-831, momentum.py:65-70
        tprv = t
        t = 0.5 * float(1.0 + np.sqrt(1.0 + 4.0 * t ** 2))
        Yf = Xf + ((tprv - 1.0) / t) * (Xf - Xfprv)
        if stats:
            rsdl = rfl2norm2(Xf - Yfprv, X.shape)   # pgm/cbpdn.py:314-320
            Ef = inner(Df, Xf, axis=AX_K) - Sf
            dfd = rfl2norm2(Ef, S.shape) / 2.0
            rl1 = np.linalg.norm((wl1 * X).ravel(), 1)
            tr['ObjFun'].append(float(dfd + lmbda * rl1))
            tr['DFid'].append(float(dfd))
            tr['RegL1'].append(float(rl1))
            tr['Rsdl'].append(float(rsdl))
            if rsdl < rel_tol:
                break
        if time_budget is not None and time.perf_counter() - t0 > time_budget:
            break
    elapsed = time.perf_counter() - t0
    out = {key: np.array(val) for key, val in tr.items()}
    out.update(X=X, Xf=Xf, Yf=Yf, Df=Df, Sf=Sf, iters=k + 1, seconds=elapsed)
    return out


# ---------------------------------------------------------------------------
# Dictionary update pieces (sporco/cnvrep.py, sporco/pgm/ccmod.py)
# ---------------------------------------------------------------------------

def zpad(v, Nv):
    """Zero-pad the spatial axes of v to Nv -- sporco/cnvrep.py:704-726."""
    vp = np.zeros(tuple(Nv) + v.shape[len(Nv):], dtype=v.dtype)
    vp[tuple(slice(0, x) for x in v.shape)] = v
    return vp


def bcrop(v, dsz):
    """Crop to the filter support (single-size dictionaries only) --
    sporco/cnvrep.py:729-795."""
    return v[:dsz[0], :dsz[1]]


def normalise(v, dimN=2):
    """Unit l2 norm per filter over spatial+channel axes; zero filters are
    left unchanged -- sporco/cnvrep.py:673-700."""
    axisN = tuple(range(0, dimN))
    vn = np.sqrt(np.sum(v ** 2, axisN, keepdims=True))
    vn[vn == 0] = 1.0
    return np.asarray(v / vn, dtype=v.dtype)


def pcn(x, dsz, Nv, dimN=2, dimC=1, crp=False, zm=False):
    """Constraint-set projection normalise(zpad(bcrop(x))) (optionally cropped
    output, optional zero-mean) -- sporco/cnvrep.py:868-913 (Pcn), :953-1074 (_Pcn*).
    zeromean (cnvrep.py:609-670) subtracts the mean over the filter support only,
    which for a single-size dictionary equals subtracting it before zero-padding."""
    v = bcrop(x, dsz)
    if zm:
        v = v - np.mean(v, axis=tuple(range(dimN)), keepdims=True)
    if not crp:
        v = zpad(v, Nv)
    return normalise(v, dimN + dimC)


# ---------------------------------------------------------------------------
# ADMM consensus dictionary update (sporco/admm/ccmod.py:605-908 on the
# ADMMConsensus base, sporco/admm/admm.py:1441-1707)
# ---------------------------------------------------------------------------

def admm_ccmod_cns(Z, S, dsz, dtype=np.float64, maxiter=20, rho=None, rlx=1.8,
                   auto_rho=False, rho_period=10, rho_tau=2.0, rho_mu=10.0,
                   rho_xi=1.0, auto_scaling=False, zero_mean=False, Y0=None,
                   abs_tol=0.0, rel_tol=1e-3):
    """ConvCnstrMOD_Consensus for a single-channel dictionary.

    ``Z``: (H, W, 1, Nb, M) coefficient maps, ``S``: (H, W, 1, Nb, 1), ``dsz`` =
    (dH, dW, M).  One dictionary copy X_i per image (the blocks of the consensus
    problem, stacked on a new last axis), consensus variable Y (H, W, 1, 1, M):

      xstep  (ccmod.py:766-778): X_i = irfftn(solvedbi_sm(Zf_i, rho,
             conj(Zf_i) Sf_i + rho rfftn(Y - U_i)))
      relax  (admm.py:1608-1616), ystep (admm.py:1585-1591 with prox_g = Pcn,
             ccmod.py:832-835), ustep (admm.py:434-437 with rsdl_r :1673-1676)
      residuals (admm.py:1673-1707), objective at Y (fEvalX False, gEvalY True:
             ccmod.py:853-894), update_rho (admm.py:549-575).
    The default rho is 1.0: the constructor's ``dval=cri.K`` (ccmod.py:700) comes
    after the base class has already set the attribute.
    """
    dtype = np.dtype(dtype)
    rdt = real_dtype(dtype).type
    Z = np.asarray(Z, dtype=dtype)
    S = np.asarray(S, dtype=dtype)
    H, W = S.shape[0], S.shape[1]
    Nb, M = Z.shape[AX_N], Z.shape[AX_K]
    rho = rdt(1.0 if rho is None else rho)
    rlx = rdt(rlx)
    Sf = rfftn2(S)
    Zf = rfftn2(Z)
    ZSf = np.conj(Zf) * Sf
    P = lambda v: pcn(v, dsz, (H, W), 2, 1, crp=False, zm=zero_mean)
    yshape = (H, W, 1, 1, M)
    if Y0 is None:
        Y = np.zeros(yshape, dtype=dtype)
        U = np.zeros(yshape + (Nb,), dtype=dtype)
    else:
        Y = np.asarray(Y0).astype(dtype, copy=True)
        U = (np.repeat(Y[..., np.newaxis], Nb, axis=-1) / rho).astype(dtype)
    Nx = Nb * int(np.prod(yshape))
    tr = {k: [] for k in ('DFid', 'Cnstr', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal',
                          'EpsDual', 'Rho')}
    X = None
    for k in range(maxiter):
        Yprev = Y.copy()
        YU = Y[..., np.newaxis] - U
        b = np.swapaxes(ZSf[..., np.newaxis], AX_N, -1) + rho * rfftn2(YU)
        Xf = np.empty_like(b)
        for i in range(Nb):
            Xf[..., i] = solvedbi_sm(Zf[..., [i], :], rho, b[..., i], None, AX_K)
        X = irfftn2(Xf, (H, W))
        AX = X if rlx == 1.0 else rlx * X + (1 - rlx) * Y[..., np.newaxis]
        Y = P(np.mean(AX + U, axis=-1)).astype(dtype)
        U = U + (AX - Y[..., np.newaxis])
        nr = np.linalg.norm(X - Y[..., np.newaxis])
        ns = np.linalg.norm(np.sqrt(Nb) * rho * (Yprev - Y))
        rn = max(np.linalg.norm(X), np.sqrt(Nb) * np.linalg.norm(Y))
        sn = rho * np.linalg.norm(U)
        rn = 1.0 if rn == 0.0 else rn
        sn = 1.0 if sn == 0.0 else sn
        r, s = nr / rn, ns / sn
        epri = np.sqrt(Nx) * abs_tol / rn + rel_tol
        edua = np.sqrt(Nx) * abs_tol / sn + rel_tol
        Ef = inner(Zf, rfftn2(Y), axis=AX_K) - Sf
        dfd = rfl2norm2(Ef, S.shape) / 2.0
        cns = np.linalg.norm(P(Y) - Y)
        for key, val in (('DFid', dfd), ('Cnstr', cns), ('PrimalRsdl', r), ('DualRsdl', s),
                         ('EpsPrimal', epri), ('EpsDual', edua), ('Rho', rho)):
            tr[key].append(float(val))
        if auto_rho and k != 0 and (k + 1) % rho_period == 0:
            if auto_scaling:
                if s == 0.0 or r == 0.0:
                    rhomlt = rho_tau
                else:
                    rhomlt = min(np.sqrt(r / (s * rho_xi) if r > s * rho_xi
                                         else (s * rho_xi) / r), rho_tau)
            else:
                rhomlt = rho_tau
            rsf = 1.0
            if r > rho_xi * rho_mu * s:
                rsf = rhomlt
            elif s > (rho_mu / rho_xi) * r:
                rsf = 1.0 / rhomlt
            rho = rho * rdt(rsf)
            U = U / rsf
        if r < epri and s < edua:
            break
    out = {key: np.array(val) for key, val in tr.items()}
    out.update(X=X, Y=Y, U=U, rho=rho, iters=k + 1, D=bcrop(Y, dsz))
    return out


# ---------------------------------------------------------------------------
# ADMM dictionary updates with a single dictionary copy: ConvCnstrMOD_IterSM and
# ConvCnstrMOD_CG (sporco/admm/ccmod.py:433-601) on ConvCnstrMODBase (:103-429)
# and ADMMEqual (sporco/admm/admm.py:808-983)
# ---------------------------------------------------------------------------

def cg_solve(matvec, b, x0, rtol, maxiter):
    """Conjugate gradients as scipy.sparse.linalg.cg (1.15) runs it for
    linalg.solvemdbi_cg (sporco/linalg.py:515-579): no preconditioner, stop when
    ||r|| < rtol ||b|| at the top of an iteration.  Returns (x, info) with scipy's
    ``info``: 0 on convergence, ``maxiter`` otherwise -- that flag is what the
    reference records as 'XSlvCGIt'."""
    x = x0.copy()
    bn = np.linalg.norm(b.ravel())
    if bn == 0:
        return b.copy(), 0
    atol = rtol * bn
    r = b - matvec(x) if x.any() else b.copy()
    p, rho_prev = None, None
    for it in range(maxiter):
        if np.linalg.norm(r.ravel()) < atol:
            return x, 0
        rho_cur = np.vdot(r.ravel(), r.ravel())
        p = r.copy() if it == 0 else r + (rho_cur / rho_prev) * p
        q = matvec(p)
        alpha = rho_cur / np.vdot(p.ravel(), q.ravel())
        x = x + alpha * p
        r = r - alpha * q
        rho_prev = rho_cur
    return x, maxiter


def admm_ccmod_eq(Z, S, dsz, method='ism', dtype=np.float64, maxiter=20, rho=None,
                  rlx=1.8, auto_rho=True, rho_period=1, rho_tau=1000.0, rho_mu=1.2,
                  rho_xi=1.0, auto_scaling=True, zero_mean=False, Y0=None,
                  aux_var_obj=False, lin_solve_check=False, cg_tol=1e-3,
                  cg_maxiter=1000, abs_tol=0.0, rel_tol=1e-3):
    """ConvCnstrMOD_IterSM (``method='ism'``) / ConvCnstrMOD_CG (``'cg'``) for a
    single-channel dictionary.

    ``Z``: (H, W, 1, Nb, M), ``S``: (H, W, 1, Nb, 1), ``dsz`` = (dH, dW, M); X, Y, U
    are (H, W, 1, 1, M).
      xstep  b = sum_n conj(Zf_n) Sf_n + rho rfftn(Y - U); Xf = (Z^H Z + rho I)^-1 b
             by iterated Sherman-Morrison (ccmod.py:496-505) or CG warm-started from
             the previous Xf (:587-601; Xf starts at 0, :583)
      relax  admm.py:877-885, ystep Y = Pcn(AX + U) (ccmod.py:363-368), ustep
             admm.py:434-437, residuals admm.py:462-486 with :959-983
      objective (ccmod.py:372-410): DFid at Xf (or rfftn(Y) with AuxVarObj), Cnstr =
             ||Pcn(v) - v|| at X (or Y)
    Default rho is 1.0 (the ``dval=cri.K`` of ccmod.py:264 comes after the base class
    has set the attribute); uinit (:298-307): U0 = Y0.
    """
    dtype = np.dtype(dtype)
    rdt = real_dtype(dtype).type
    Z = np.asarray(Z, dtype=dtype)
    S = np.asarray(S, dtype=dtype)
    H, W = S.shape[0], S.shape[1]
    M = Z.shape[AX_K]
    rho = rdt(1.0 if rho is None else rho)
    rlx = rdt(rlx)
    Sf = rfftn2(S)
    Zf = rfftn2(Z)
    ZSf = inner(np.conj(Zf), Sf, axis=AX_N)
    P = lambda v: pcn(v, dsz, (H, W), 2, 1, crp=False, zm=zero_mean)
    yshape = (H, W, 1, 1, M)
    if Y0 is None:
        Y = np.zeros(yshape, dtype=dtype)
        U = np.zeros(yshape, dtype=dtype)
    else:
        Y = np.asarray(Y0).astype(dtype, copy=True)
        U = Y.copy()
    Xf = np.zeros((H, W // 2 + 1, 1, 1, M), dtype=complex_dtype(dtype))
    Nx = int(np.prod(yshape))
    AHA = lambda x: inner(np.conj(Zf), inner(Zf, x, axis=AX_K), axis=AX_N)
    keys = ('DFid', 'Cnstr', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho',
            'XSlvRelRes') + (('XSlvCGIt',) if method == 'cg' else ())
    tr = {k: [] for k in keys}
    X = None
    for k in range(maxiter):
        Yprev = Y.copy()
        b = ZSf + rho * rfftn2(Y - U)
        if method == 'ism':
            Xf = solvemdbi_ism(Zf, rho, b, AX_K, AX_N)
            cgit = None
        else:
            Xf, cgit = cg_solve(lambda x: AHA(x) + rho * x, b, Xf, cg_tol, cg_maxiter)
        Xf = Xf.astype(complex_dtype(dtype))
        X = irfftn2(Xf, (H, W))
        xrrs = rrs(AHA(Xf) + rho * Xf, b) if lin_solve_check else np.nan
        AX = X if rlx == 1.0 else rlx * X + (1 - rlx) * Y
        Y = P(AX + U).astype(dtype)
        U = U + (AX - Y)
        nr = np.linalg.norm(X - Y)
        ns = rho * np.linalg.norm(Yprev - Y)
        rn = max(np.linalg.norm(X), np.linalg.norm(Y))
        sn = rho * np.linalg.norm(U)
        rn = 1.0 if rn == 0.0 else rn
        sn = 1.0 if sn == 0.0 else sn
        r, s = nr / rn, ns / sn
        epri = np.sqrt(Nx) * abs_tol / rn + rel_tol
        edua = np.sqrt(Nx) * abs_tol / sn + rel_tol
        fv = rfftn2(Y) if aux_var_obj else Xf
        gv = Y if aux_var_obj else X
        dfd = rfl2norm2(inner(Zf, fv, axis=AX_K) - Sf, S.shape) / 2.0
        cns = np.linalg.norm(P(gv) - gv)
        vals = dict(DFid=dfd, Cnstr=cns, PrimalRsdl=r, DualRsdl=s, EpsPrimal=epri,
                    EpsDual=edua, Rho=rho, XSlvRelRes=xrrs, XSlvCGIt=cgit)
        for key in keys:
            tr[key].append(float(vals[key]))
        if auto_rho and k != 0 and (k + 1) % rho_period == 0:
            if auto_scaling:
                if s == 0.0 or r == 0.0:
                    rhomlt = rho_tau
                else:
                    rhomlt = min(np.sqrt(r / (s * rho_xi) if r > s * rho_xi
                                         else (s * rho_xi) / r), rho_tau)
            else:
                rhomlt = rho_tau
            rsf = 1.0
            if r > rho_xi * rho_mu * s:
                rsf = rhomlt
            elif s > (rho_mu / rho_xi) * r:
                rsf = 1.0 / rhomlt
            rho = rho * rdt(rsf)
            U = U / rsf
        if r < epri and s < edua:
            break
    out = {key: np.array(val) for key, val in tr.items()}
    out.update(X=X, Y=Y, U=U, rho=rho, iters=k + 1, D=bcrop(Y, dsz))
    return out


# ---------------------------------------------------------------------------
# online dictionary learning (sporco/dictlrn/onlinecdl.py:33-460)
# ---------------------------------------------------------------------------

def online_cdl(D0, batches, lmbda, dtype=np.float64, eta_a=10.0, eta_b=5.0,
               zero_mean=False, xstep_iter=100, masks=None):
    """OnlineConvBPDNDictLearn: for training batch j (5-D ``(H, W, 1, N, 1)``):
      xstep  ADMM ConvBPDN with the current dictionary, cold start, ``xstep_iter``
             iterations at the class's X-step defaults (AutoRho period 10, fixed
             scaling 2, ratio 10; onlinecdl.py:89-101, :267-287)
      dstep  gradf = sum_n conj(Zf_n) (sum_m Zf_nm Df_m - Sf_n); eta = a / (j + b);
             G = irfftn(Df - eta gradf); D = Pcn(G) cropped (:310-333)
      stats  Cnstr = ||zpad(D) - G||, DeltaD = ||D - Dprv|| (:398-399).
    With ``masks`` (one per batch, broadcastable to the batch): OnlineConvBPDNMaskDictLearn
    (:464-600) -- X-step ConvBPDNMaskDcpl at its defaults (rho 1, AutoRho off), residual of
    the dictionary gradient weighted by W ONCE in the spatial domain (:578-580).
    ``D0``: (dH, dW, M)."""
    dtype = np.dtype(dtype)
    dsz = D0.shape
    M = dsz[-1]
    D = pcn(np.asarray(D0, dtype=dtype).reshape(dsz[0], dsz[1], 1, 1, M), dsz, (), 2, 1,
            crp=True, zm=zero_mean)
    tr = {k: [] for k in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho', 'Cnstr',
                          'DeltaD', 'Eta')}
    Ds = []
    for j, S in enumerate(batches):
        S = np.asarray(S, dtype=dtype)
        H, W = S.shape[:2]
        if masks is None:
            r = admm_cbpdn(D, S, lmbda, dtype=dtype, maxiter=xstep_iter, rho_period=10,
                           rho_tau=2.0, rho_mu=10.0, rho_xi=1.0, auto_scaling=False)
            Zf = rfftn2(r['Y'])
        else:
            Wm = np.asarray(masks[j], dtype=dtype)
            r = admm_cbpdn_maskdcpl(D, S, lmbda, Wm, dtype=dtype, maxiter=xstep_iter)
            Zf = rfftn2(r['Y1'])
        Sf = rfftn2(S)
        Df = rfftn2(D, (H, W))
        Ryf = inner(Zf, Df, axis=AX_K) - Sf
        if masks is not None:
            Ryf = rfftn2(Wm * irfftn2(Ryf, (H, W)))
        gradf = inner(np.conj(Zf), Ryf, axis=AX_N)
        eta = eta_a / (j + eta_b)
        G = irfftn2(Df - eta * gradf, (H, W))
        Dprv = D
        D = pcn(G, dsz, (H, W), 2, 1, crp=True, zm=zero_mean).astype(dtype)
        vals = dict(ObjFun=r['ObjFun'][-1], DFid=r['DFid'][-1], RegL1=r['RegL1'][-1],
                    PrimalRsdl=r['PrimalRsdl'][-1], DualRsdl=r['DualRsdl'][-1],
                    Rho=r['Rho'][-1], Cnstr=np.linalg.norm(zpad(D, (H, W)) - G),
                    DeltaD=np.linalg.norm(D - Dprv), Eta=eta)
        for k, v in vals.items():
            tr[k].append(float(v))
        Ds.append(D)
    out = {k: np.array(v) for k, v in tr.items()}
    out['Ds'] = np.stack(Ds)
    return out


# ---------------------------------------------------------------------------
# ConvBPDNMaskDcpl (sporco/admm/cbpdn.py:2066-2283) on ConvTwoBlockCnstrnt
# (:1401-1826) and ADMMTwoBlockCnstrnt (sporco/admm/admm.py:989-1437)
# ---------------------------------------------------------------------------

def admm_cbpdn_maskdcpl(D, S, lmbda, W, dtype=np.float64, maxiter=50, rho=1.0, rlx=1.8,
                        auto_rho=False, rho_period=10, rho_tau=2.0, rho_mu=10.0,
                        rho_xi=1.0, auto_scaling=False, abs_tol=0.0, rel_tol=1e-3,
                        nonneg=False, nobndry=False, wl1=1.0, aux_var_obj=False,
                        lin_solve_check=False):
    """Mask decoupling: minimise (1/2)||W(sum_m d_m * x_m - s)||^2 + lmbda ||x||_1 with
    the constraint [D; I] x - [y0; y1] = [s; 0], single-channel dictionary.

    ``D``: (dH, dW, 1, 1, K); ``S``: (H, W, C, N, 1); ``W`` broadcastable to S.
      xstep  (cbpdn.py:1610-1643): b = conj(Df) rfftn(y0 - u0 + s) + rfftn(y1 - u1);
             Xf = solvedbi_sm(Df, 1.0, b) -- rho does not enter
      relax  (:1664-1677): AXnr = [D x; x], AX = a AXnr + (1 - a) [y0 + s; y1]
      ystep  (:2236-2247, :1647-1660): y0 = rho (AX0 + u0 - s) / (W^2 + rho),
             y1 = prox_l1(AX1 + u1, (lmbda / rho) wl1) (+ NonNegCoef / NoBndryCross)
      ustep  (admm.py:434-437 with rsdl_r :1404-1414): u += AX - [y0 + s; y1]
      residuals (admm.py:462-486): r = ||AXnr - [y0 + s; y1]||, s = rho ||A^T u|| with
             A^T u = irfftn(conj(Df) rfftn(u0)) + u1 (cbpdn.py:1814-1818, the NEW u),
             rn = max(||AXnr||, ||y||, ||s||) (admm.py:1423-1431), sn = rho ||u||
             (cbpdn.py:1821-1824); Nx = K H W N, Nc = size of y (cbpdn.py:1568-1574)
      objective (cbpdn.py:2251-2275): DFid = (1/2)||W g0||^2, g0 = y0 (AuxVarObj) or
             D x - s; RegL1 = ||wl1 g1||_1, g1 = y1 or x.
    Returns Y1 as the coefficient maps (ReturnVar 'Y1', :1487)."""
    dtype = np.dtype(dtype)
    rdt = real_dtype(dtype).type
    D = np.asarray(D, dtype=dtype)
    S = np.asarray(S, dtype=dtype)
    W = np.asarray(W, dtype=dtype)
    H, Wd = S.shape[0], S.shape[1]
    K = D.shape[AX_K]
    shpX = (H, Wd, S.shape[AX_C], S.shape[AX_N], K)
    Nx = K * H * Wd * S.shape[AX_N]
    Nc = int(np.prod(shpX)) + int(np.prod(S.shape))
    lmbda, rho, rlx = rdt(lmbda), rdt(rho), rdt(rlx)
    wl1 = np.asarray(wl1, dtype=real_dtype(dtype))
    Df = rfftn2(D, (H, Wd))
    A0 = lambda xf: irfftn2(inner(Df, xf, axis=AX_K), (H, Wd))
    A0T = lambda y0: irfftn2(np.conj(Df) * rfftn2(y0), (H, Wd))
    Y0 = np.zeros(S.shape, dtype=dtype)
    Y1 = np.zeros(shpX, dtype=dtype)
    U0, U1 = Y0.copy(), Y1.copy()
    nrm_c = np.linalg.norm(S)
    nrm2 = lambda a, b: np.sqrt(np.linalg.norm(a) ** 2 + np.linalg.norm(b) ** 2)
    tr = {k: [] for k in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal',
                          'EpsDual', 'Rho', 'XSlvRelRes')}
    X = None
    for k in range(maxiter):
        b = np.conj(Df) * rfftn2(Y0 - U0 + S) + rfftn2(Y1 - U1)
        Xf = solvedbi_sm(Df, 1.0, b, None, AX_K).astype(b.dtype)
        X = irfftn2(Xf, (H, Wd))
        xrrs = rrs(np.conj(Df) * inner(Df, Xf, axis=AX_K) + Xf, b) if lin_solve_check \
            else np.nan
        AX0nr, AX1nr = A0(Xf), X
        if rlx == 1.0:
            AX0, AX1 = AX0nr, AX1nr
        else:
            AX0 = rlx * AX0nr + (1 - rlx) * (Y0 + S)
            AX1 = rlx * AX1nr + (1 - rlx) * Y1
        Y0 = ((rho * (AX0 + U0 - S)) / (W ** 2 + rho)).astype(dtype)
        Y1 = prox_l1(AX1 + U1, (lmbda / rho) * wl1).astype(dtype)
        if nonneg:
            Y1[Y1 < 0.0] = 0.0
        if nobndry:
            Y1[1 - D.shape[0]:] = 0.0
            Y1[:, 1 - D.shape[1]:] = 0.0
        U0 = U0 + (AX0 - (Y0 + S))
        U1 = U1 + (AX1 - Y1)
        nr = nrm2(AX0nr - (Y0 + S), AX1nr - Y1)
        ns = rho * np.linalg.norm(A0T(U0) + U1)
        rn = max(nrm2(AX0nr, AX1nr), nrm2(Y0, Y1), nrm_c)
        sn = rho * nrm2(U0, U1)
        rn = 1.0 if rn == 0.0 else rn
        sn = 1.0 if sn == 0.0 else sn
        r, s = nr / rn, ns / sn
        epri = np.sqrt(Nc) * abs_tol / rn + rel_tol
        edua = np.sqrt(Nx) * abs_tol / sn + rel_tol
        g0 = Y0 if aux_var_obj else AX0nr - S
        g1 = Y1 if aux_var_obj else X
        dfd = np.linalg.norm(W * g0) ** 2 / 2.0
        rl1 = np.sum(np.abs(wl1 * g1))
        vals = dict(ObjFun=dfd + lmbda * rl1, DFid=dfd, RegL1=rl1, PrimalRsdl=r, DualRsdl=s,
                    EpsPrimal=epri, EpsDual=edua, Rho=rho, XSlvRelRes=xrrs)
        for key, val in vals.items():
            tr[key].append(float(val))
        if auto_rho and k != 0 and (k + 1) % rho_period == 0:
            if auto_scaling:
                if s == 0.0 or r == 0.0:
                    rhomlt = rho_tau
                else:
                    rhomlt = min(np.sqrt(r / (s * rho_xi) if r > s * rho_xi
                                         else (s * rho_xi) / r), rho_tau)
            else:
                rhomlt = rho_tau
            rsf = 1.0
            if r > rho_xi * rho_mu * s:
                rsf = rhomlt
            elif s > (rho_mu / rho_xi) * r:
                rsf = 1.0 / rhomlt
            rho = rho * rdt(rsf)
            U0, U1 = U0 / rsf, U1 / rsf
        if r < epri and s < edua:
            break
    out = {key: np.array(val) for key, val in tr.items()}
    out.update(X=X, Y0=Y0, Y1=Y1, U0=U0, U1=U1, rho=rho, iters=k + 1)
    return out


# ---------------------------------------------------------------------------
# Dictionary updates with mask decoupling: ConvCnstrMODMaskDcpl_IterSM / _CG
# (sporco/admm/ccmodmd.py:27-762) on ADMMTwoBlockCnstrnt
# ---------------------------------------------------------------------------

def admm_ccmod_maskdcpl(Z, S, W, dsz, method='ism', dtype=np.float64, maxiter=20, rho=1.0,
                        rlx=1.8, auto_rho=False, rho_period=10, rho_tau=2.0, rho_mu=10.0,
                        rho_xi=1.0, auto_scaling=False, zero_mean=False, Y0=None,
                        aux_var_obj=False, lin_solve_check=False, cg_tol=1e-3,
                        cg_maxiter=1000, abs_tol=0.0, rel_tol=1e-3):
    """Constraint [Z; I] d - [y0; y1] = [s; 0], single-channel dictionary.

    ``Z``: (H, W, 1, Nb, M), ``S``: (H, W, 1, Nb, 1), ``W`` broadcastable to S, ``dsz`` =
    (dH, dW, M); d, y1, u1 are (H, W, 1, 1, M), y0, u0 have the shape of S.
      xstep  (ccmodmd.py:638-654 / :735-753): b = sum_n conj(Zf_n) rfftn(y0 - u0 + s)_n +
             rfftn(y1 - u1); (Z^H Z + I) Xf = b by iterated Sherman-Morrison or CG warm
             started from the previous Xf -- rho does not enter
      relax  (:387-396), ystep (:374-383): y0 = rho (AX0 + u0 - s) / (W^2 + rho),
             y1 = Pcn(AX1 + u1); ustep admm.py:434-437 with rsdl_r :1404-1414
      residuals: r = ||AXnr - [y0 + s; y1]||, s = rho ||A^T u|| with
             A^T u = irfftn(sum_n conj(Zf_n) rfftn(u0)_n) + u1 (:557-561), rn =
             max(||AXnr||, ||y||, ||s||), sn = rho ||u|| (:564-567); Nx = size of d,
             Nc = size of y (:262-270)
      objective (:508-532): DFid = (1/2)||W g0||^2, g0 = y0 (AuxVarObj) or Z d - s;
             Cnstr = ||Pcn(g1) - g1||, g1 = y1 or d.
    ``Y0`` (block 1 only, block 0 zero, as dictionary learning passes it,
    cbpdndlmd.py:436-444): y1 = u1 = Y0 (uinit :311-322)."""
    dtype = np.dtype(dtype)
    rdt = real_dtype(dtype).type
    Z = np.asarray(Z, dtype=dtype)
    S = np.asarray(S, dtype=dtype)
    W = np.asarray(W, dtype=dtype)
    H, Wd = S.shape[0], S.shape[1]
    M = Z.shape[AX_K]
    rho, rlx = rdt(rho), rdt(rlx)
    Zf = rfftn2(Z)
    P = lambda v: pcn(v, dsz, (H, Wd), 2, 1, crp=False, zm=zero_mean)
    A0 = lambda xf: irfftn2(inner(Zf, xf, axis=AX_K), (H, Wd))
    A0T = lambda y0: irfftn2(inner(np.conj(Zf), rfftn2(y0), axis=AX_N), (H, Wd))
    AHA = lambda x: inner(np.conj(Zf), inner(Zf, x, axis=AX_K), axis=AX_N)
    yshape = (H, Wd, 1, 1, M)
    Y0b = np.zeros(S.shape, dtype=dtype)
    U0b = Y0b.copy()
    if Y0 is None:
        Y1 = np.zeros(yshape, dtype=dtype)
        U1 = Y1.copy()
    else:
        Y1 = np.asarray(Y0).astype(dtype, copy=True)
        U1 = Y1.copy()
    Xf = np.zeros((H, Wd // 2 + 1, 1, 1, M), dtype=complex_dtype(dtype))
    Nx = int(np.prod(yshape))
    Nc = Nx + int(np.prod(S.shape))
    nrm_c = np.linalg.norm(S)
    nrm2 = lambda a, b: np.sqrt(np.linalg.norm(a) ** 2 + np.linalg.norm(b) ** 2)
    keys = ('DFid', 'Cnstr', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho',
            'XSlvRelRes') + (('XSlvCGIt',) if method == 'cg' else ())
    tr = {k: [] for k in keys}
    X = None
    for k in range(maxiter):
        b = inner(np.conj(Zf), rfftn2(Y0b - U0b + S), axis=AX_N) + rfftn2(Y1 - U1)
        if method == 'ism':
            Xf = solvemdbi_ism(Zf, 1.0, b, AX_K, AX_N)
            cgit = None
        else:
            Xf, cgit = cg_solve(lambda x: AHA(x) + x, b, Xf, cg_tol, cg_maxiter)
        Xf = Xf.astype(complex_dtype(dtype))
        X = irfftn2(Xf, (H, Wd))
        xrrs = rrs(AHA(Xf) + Xf, b) if lin_solve_check else np.nan
        AX0nr, AX1nr = A0(Xf), X
        if rlx == 1.0:
            AX0, AX1 = AX0nr, AX1nr
        else:
            AX0 = rlx * AX0nr + (1 - rlx) * (Y0b + S)
            AX1 = rlx * AX1nr + (1 - rlx) * Y1
        Y0b = ((rho * (AX0 + U0b - S)) / (W ** 2 + rho)).astype(dtype)
        Y1 = P(AX1 + U1).astype(dtype)
        U0b = U0b + (AX0 - (Y0b + S))
        U1 = U1 + (AX1 - Y1)
        nr = nrm2(AX0nr - (Y0b + S), AX1nr - Y1)
        ns = rho * np.linalg.norm(A0T(U0b) + U1)
        rn = max(nrm2(AX0nr, AX1nr), nrm2(Y0b, Y1), nrm_c)
        sn = rho * nrm2(U0b, U1)
        rn = 1.0 if rn == 0.0 else rn
        sn = 1.0 if sn == 0.0 else sn
        r, s = nr / rn, ns / sn
        epri = np.sqrt(Nc) * abs_tol / rn + rel_tol
        edua = np.sqrt(Nx) * abs_tol / sn + rel_tol
        g0 = Y0b if aux_var_obj else AX0nr - S
        g1 = Y1 if aux_var_obj else X
        vals = dict(DFid=np.linalg.norm(W * g0) ** 2 / 2.0, Cnstr=np.linalg.norm(P(g1) - g1),
                    PrimalRsdl=r, DualRsdl=s, EpsPrimal=epri, EpsDual=edua, Rho=rho,
                    XSlvRelRes=xrrs, XSlvCGIt=cgit)
        for key in keys:
            tr[key].append(float(vals[key]))
        if auto_rho and k != 0 and (k + 1) % rho_period == 0:
            if auto_scaling:
                if s == 0.0 or r == 0.0:
                    rhomlt = rho_tau
                else:
                    rhomlt = min(np.sqrt(r / (s * rho_xi) if r > s * rho_xi
                                         else (s * rho_xi) / r), rho_tau)
            else:
                rhomlt = rho_tau
            rsf = 1.0
            if r > rho_xi * rho_mu * s:
                rsf = rhomlt
            elif s > (rho_mu / rho_xi) * r:
                rsf = 1.0 / rhomlt
            rho = rho * rdt(rsf)
            U0b, U1 = U0b / rsf, U1 / rsf
        if r < epri and s < edua:
            break
    out = {key: np.array(val) for key, val in tr.items()}
    out.update(X=X, Y0=Y0b, Y1=Y1, U0=U0b, U1=U1, rho=rho, iters=k + 1, D=bcrop(Y1, dsz))
    return out
