"""Image sizes 16 x {10, 12, 14, 15, 18, 20, 21, 24, 25, 27, 28, 30} = 160 ... 480 on the register-resident
kernels (round 6: csc_rows_mr.hip, csc_fused.h; in-register transforms of that many points,
radices 7 / 5 / 3 / 2, regfft.h) -- the sizes the
reference serves in the same speed class as any other (sporco/fft.py:257-314; its own tests are
odd-sized: tests/admm/test_cbpdn.py:204-225).

* against runs of the UNMODIFIED reference at the shapes of the bench line (tests/golden/admm_mr_*,
  written by oracle/make_golden.py mixed_radix), with the kernel counters asserting that the
  mixed-radix instantiations ran;
* against the float64 oracle and the generic chain of this library at small filter counts (what the
  CPU simulator finishes in a minute);
* the single-array state bit for bit against the (Y, U) form;
* every option set the mixed-radix kernels do not serve takes the generic chain of the same handle
  and gives the generic chain's results."""

import os

import numpy as np
import pytest

from conftest import load_golden, rel_l2

TRACES = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')


def problem(H, W, K, N, seed):
    rng = np.random.RandomState(seed)
    D = rng.randn(4, 4, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    return D, rng.randn(H, W, N).astype(np.float32)


class env(object):
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k in self.kv:
            os.environ.pop(k, None)


def kernel_counts(b):
    return {k: v[1] for k, v in b._dev.profile_read().items() if v[1] > 0}


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['admm_mr_384x384_k32_n2', 'admm_mr_480x320_k64_n1',
                                  'admm_mr_448x384_k8_nonneg_n2', 'admm_mr_240x320_k64_n2'])
def test_reference_fixtures_at_mixed_radix_shapes(gpu_backend, name):
    """float32 on the register kernels against the reference's float64 run of the same inputs:
    coefficient maps within the BASELINE bar (1e-4; measured ~1e-6), traces 1e-3 (measured ~1e-6);
    device-driven loop and host-driven loop (whose counters show the kernels that ran)."""
    import bench
    from sporco_amd.admm import cbpdn
    g = load_golden(name)
    H, W, K, N = [int(v) for v in g['shape']]
    D, S = bench.make_problem(H, W, K, N, 0)
    optd = {'MaxMainIter': len(g['it_Iter']), 'RelStopTol': 0.0}
    if 'nonneg' in name:
        optd.update({'NonNegCoef': True, 'AutoRho': {'Period': 2}})
    for host in (False, True):
        with env(**({'SPORCO_AMD_HOST_LOOP': '1'} if host else {})):
            b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
            assert b._dev.uses_fused_rows() and b._dev.uses_fused_cols() and b._fused_ok()
            assert host or b._device_loop_ok()
            b._dev.profile(True)
            Y = b.solve()
            kc = kernel_counts(b)
        Y = Y.reshape(H, W, 1, N, K)
        assert rel_l2(Y[::8, ::8], g['Y_sub']) < 1e-4
        assert abs(np.linalg.norm(Y.astype(np.float64)) / float(g['Y_l2']) - 1.0) < 1e-5
        its = b.getitstat()
        for f in TRACES:
            assert rel_l2(getattr(its, f), g['it_' + f]) < 1e-3, f
        if host:
            # the three-launch iteration in the single-array state: no generic-chain kernel ran
            assert kc.get('fused_cols_sm', 0) == len(g['it_Iter'])
            assert kc.get('rows_inv_post_v', 0) + kc.get('rows_inv_post_v_emit', 0) >= len(g['it_Iter']) - 2
            assert not any(k.startswith(('fft_', 'sm_solve', 'admm_post')) for k in kc), kc


@pytest.mark.parametrize('H,W,K,N', [pytest.param(384, 320, 4, 1, marks=pytest.mark.gpu),
                                     # odd points per thread (15), one exchange group (10 <= 16)
                                     (240, 160, 4, 1), (160, 336, 4, 1),
                                     # K > 64: cooperating 64-filter slab workgroups at a mixed-radix height
                                     (240, 160, 66, 1),
                                     pytest.param(384, 320, 128, 1, marks=pytest.mark.gpu),
                                     pytest.param(480, 240, 96, 2, marks=pytest.mark.gpu),
                                     pytest.param(448, 336, 70, 1, marks=pytest.mark.gpu),
                                     pytest.param(512, 400, 128, 1, marks=pytest.mark.gpu),
                                     pytest.param(240, 320, 64, 2, marks=pytest.mark.gpu),
                                     pytest.param(224, 224, 14, 3, marks=pytest.mark.gpu),
                                     pytest.param(336, 400, 8, 1, marks=pytest.mark.gpu),
                                     pytest.param(432, 288, 6, 2, marks=pytest.mark.gpu),
                                     pytest.param(160, 192, 64, 2, marks=pytest.mark.gpu),
                                     pytest.param(512, 240, 10, 1, marks=pytest.mark.gpu),
                                     pytest.param(480, 448, 6, 1, marks=pytest.mark.gpu),
                                     pytest.param(448, 480, 64, 2, marks=pytest.mark.gpu),
                                     pytest.param(320, 512, 8, 2, marks=pytest.mark.gpu),
                                     pytest.param(128, 384, 6, 3, marks=pytest.mark.gpu),
                                     pytest.param(384, 256, 62, 1, marks=pytest.mark.gpu),
                                     pytest.param(320, 320, 7, 2, marks=pytest.mark.gpu),
                                     # three and four cooperating slabs (K = 192, 256), the minimal K
                                     pytest.param(320, 240, 192, 1, marks=pytest.mark.gpu),
                                     pytest.param(240, 400, 256, 1, marks=pytest.mark.gpu),
                                     pytest.param(160, 480, 2, 1, marks=pytest.mark.gpu)])
def test_mixed_radix_sizes_vs_oracle_and_generic_chain(backend, H, W, K, N):
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    D, S = problem(H, W, K, N, seed=H + W + K)
    # (the float64 oracle runs on the host: fewer iterations for the big cases)
    iters = 4 if backend == 'hostsim' else (12 if H * W * K * N <= 4e6 else 3)
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0}
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert b._dev.uses_fused_rows() and b._dev.uses_fused_cols() and b._device_loop_ok()
    Y = b.solve()
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05, dtype=np.float64,
                         maxiter=iters, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 2e-5 and rel_l2(b.U, ref['U']) < 2e-5
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 2e-5, f
    # X never left the registers: rebuilt on demand, and Xf from X
    assert rel_l2(b.X, ref['X']) < 2e-5
    assert rel_l2(b.Xf, np.fft.rfftn(ref['X'], axes=(0, 1))) < 2e-5
    if backend == 'hostsim':
        return
    with env(SPORCO_AMD_UNFUSED='1'):
        b0 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert not b0._dev.uses_fused_rows()
    assert rel_l2(Y, b0.solve()) < 2e-5
    # the solver keeps going from where it stopped (admm.py:331), now under the host-driven loop
    with env(SPORCO_AMD_HOST_LOOP='1'):
        b.solve()
    ref2 = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05, dtype=np.float64,
                          maxiter=2 * iters, rel_tol=0.0)
    assert rel_l2(b.Y, ref2['Y']) < 5e-5


@pytest.mark.parametrize('case', ['default', pytest.param('nonneg_fixed_rho', marks=pytest.mark.gpu)])
def test_single_array_state_is_bit_identical_at_mixed_radix_sizes(backend, case):
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    H, W, K, N = (192, 240, 4, 1) if backend == 'hostsim' else (480, 384, 16, 2)
    D, S = problem(H, W, K, N, seed=99)
    optd = {'MaxMainIter': 5 if backend == 'hostsim' else 9, 'RelStopTol': 0.0}
    if case != 'default':
        optd.update({'NonNegCoef': True, 'AutoRho': {'Enabled': False}, 'rho': 3.0})
    outs = []
    for vform in (False, True):
        with env(**({} if vform else {'SPORCO_AMD_NO_VFORM': '1'})):
            b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
            b._return_min = False
            b.solve()
            assert b._dev.query(_lib.QUERY_VFORM_LIVE) == (1 if vform else 0)
            outs.append((b.Y.copy(), b.U.copy(), b.X.copy(),
                         {f: np.asarray(getattr(b.getitstat(), f), float) for f in TRACES}))
    for a, c in zip(outs[0][:3], outs[1][:3]):
        assert np.array_equal(a, c)
    for f in TRACES:
        assert np.array_equal(outs[0][3][f], outs[1][3][f]), f


def test_weights_and_boundary_options_on_the_mixed_radix_kernels(backend):
    """An L1Weight array (incl. a per-filter weight with a zero, the lowpass-filter idiom of the
    reference's examples), NoBndryCross and both with NonNegCoef: the MODE 1 instantiations of the
    mixed-radix row kernels (csc_rows_mr.hip) against the float64 oracle and the generic chain."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    H, W, K, N = (160, 192, 4, 1) if backend == 'hostsim' else (384, 480, 8, 2)
    D, S = problem(H, W, K, N, seed=17)
    rng = np.random.RandomState(5)
    wfull = (np.abs(rng.randn(H, W, 1, N, K)) + 0.5).astype(np.float32)
    wfilt = np.ones((1, 1, 1, 1, K), np.float32)
    wfilt[..., 0] = 0.0
    iters = 4 if backend == 'hostsim' else 10
    cases = [('filter_weight', {'L1Weight': wfilt}, dict(wl1=wfilt.astype(np.float64))),
             ('nobndry', {'NoBndryCross': True}, dict(nobndry=True))]
    if backend != 'hostsim':
        cases += [('full_weight_nonneg', {'L1Weight': wfull, 'NonNegCoef': True},
                   dict(wl1=wfull.astype(np.float64), nonneg=True)),
                  ('weight_nobndry', {'L1Weight': wfilt, 'NoBndryCross': True},
                   dict(wl1=wfilt.astype(np.float64), nobndry=True))]
    for name, extra, okw in cases:
        optd = dict({'MaxMainIter': iters, 'RelStopTol': 0.0}, **extra)
        b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
        assert b._dev.uses_fused_rows() and b._fused_ok() and b._device_loop_ok(), name
        b._dev.profile(True)
        Y = b.solve()
        kc = kernel_counts(b)
        assert not any(k.startswith(('fft_', 'sm_solve', 'admm_post')) for k in kc), (name, kc)
        ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05, dtype=np.float64,
                             maxiter=iters, rel_tol=0.0, **okw)
        assert rel_l2(Y, ref['Y']) < 2e-5 and rel_l2(b.U, ref['U']) < 2e-5 and rel_l2(b.X, ref['X']) < 2e-5, name
        its = b.getitstat()
        for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            assert rel_l2(getattr(its, f), ref[f]) < 2e-5, (name, f)


def test_other_options_take_the_generic_chain(backend):
    """LinSolveCheck (and multi-channel dictionaries, the consensus dictionary update) at a
    mixed-radix size: served by the generic chain of the handle -- the results are those of a handle
    that never had the register kernels (SPORCO_AMD_UNFUSED=1), bit for bit; the staged step methods
    run their X-step on the register kernels, FISTA its whole iteration (tested above)."""
    from sporco_amd.admm import cbpdn
    from sporco_amd.pgm import cbpdn as pc
    H, W, K, N = (160, 160, 4, 1) if backend == 'hostsim' else (384, 480, 8, 2)
    D, S = problem(H, W, K, N, seed=7)
    o = {'MaxMainIter': 3, 'RelStopTol': 0.0}

    class Hooked(cbpdn.ConvBPDN):
        def ystep(self):
            super(Hooked, self).ystep()

    cases = [
        ('staged', lambda: Hooked(D, S, 0.05, cbpdn.ConvBPDN.Options(o))),
        ('LinSolveCheck', lambda: cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(dict(o, LinSolveCheck=True)))),
    ]
    for name, make in cases:
        b = make()
        a = b.solve()
        with env(SPORCO_AMD_UNFUSED='1'):
            b0 = make()
        c = b0.solve()
        t, t0 = b.getitstat(), b0.getitstat()
        if name == 'staged':
            # (an overridden step method: one device call per step -- the X-step alone still runs
            # on the mixed-radix kernels, rows_fwd / fused_cols / the prox-less inverse row pass)
            assert rel_l2(a, c) < 2e-5 and rel_l2(t.ObjFun, t0.ObjFun) < 1e-5
            continue
        assert np.array_equal(a, c), name
        assert np.array_equal(np.asarray(t.ObjFun), np.asarray(t0.ObjFun)), name


@pytest.mark.parametrize('H,W,K,N', [(240, 160, 4, 2),
                                     pytest.param(320, 192, 4, 1, marks=pytest.mark.gpu),
                                     pytest.param(480, 384, 64, 2, marks=pytest.mark.gpu),
                                     pytest.param(336, 448, 10, 2, marks=pytest.mark.gpu)])
def test_fista_and_dictionary_learning_at_mixed_radix_sizes(backend, H, W, K, N):
    """pgm.cbpdn.ConvBPDN (fixed L, BacktrackStandard, an L1Weight array) and ConvBPDNDictLearn
    (ADMM X-step, PGM D-step) on the mixed-radix column kernels of csc_pgm_mr.hip
    (sporco/pgm/cbpdn.py:263-372, sporco/pgm/ccmod.py:295-323) against the generic chain of the
    same library -- which the reference fixtures of tests/test_pgm_cbpdn.py / test_dictlearn.py
    pin -- and the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.pgm import cbpdn as pc
    from sporco_amd.pgm.backtrack import BacktrackStandard
    from sporco_amd.dictlrn import cbpdndl
    D, S = problem(H, W, K, N, seed=H + K)
    rng = np.random.RandomState(11)
    wfilt = np.ones((1, 1, 1, 1, K), np.float32)
    wfilt[..., 0] = 0.0
    iters = 4 if backend == 'hostsim' else 12
    cases = [('backtrack', {'L': 5.0, 'Backtrack': BacktrackStandard()})]
    if backend != 'hostsim':       # (the simulator run is kept to the trial form: it exercises every kernel)
        cases += [('fixed_L', {'L': 50.0}), ('weights', {'L': 50.0, 'L1Weight': wfilt})]
    for name, extra in cases:
        runs = []
        for unfused in (False, True):
            with env(**({'SPORCO_AMD_UNFUSED': '1'} if unfused else {})):
                b = pc.ConvBPDN(D, S, 0.05, pc.ConvBPDN.Options(dict({'MaxMainIter': iters, 'RelStopTol': 0.0}, **extra)))
            assert bool(b.dev.uses_fused_pgm()) == (not unfused) and b._fused_ok() == (not unfused), name
            runs.append((b.solve(), b.getitstat()))
        assert rel_l2(runs[0][0], runs[1][0]) < 2e-5, name
        for f in ('ObjFun', 'DFid', 'RegL1', 'Rsdl', 'L'):
            assert rel_l2(getattr(runs[0][1], f), getattr(runs[1][1], f)) < 2e-5, (name, f)
        if name == 'fixed_L':
            ref = orc.pgm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05, L=50.0, dtype=np.float64,
                                maxiter=iters, rel_tol=0.0)
            assert rel_l2(runs[0][0], ref['X'].reshape(runs[0][0].shape)) < 2e-5
    D0 = rng.randn(4, 4, K).astype(np.float32)
    outs = []
    for unfused in (False, True):
        with env(**({'SPORCO_AMD_UNFUSED': '1'} if unfused else {})):
            opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 3, 'AccurateDFid': True, 'CCMOD': {'ZeroMean': True}},
                                                    xmethod='admm', dmethod='pgm')
            d = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod='pgm')
        assert bool(d.xstep._dev.uses_fused_rows()) == (not unfused)
        outs.append((d.solve(), d.getcoef(), d.getitstat()))
    assert rel_l2(outs[0][0], outs[1][0]) < 1e-5 and rel_l2(outs[0][1], outs[1][1]) < 2e-5
    for f in ('ObjFun', 'DFid', 'RegL1', 'Cnstr'):
        a, c = np.asarray(getattr(outs[0][2], f), float), np.asarray(getattr(outs[1][2], f), float)
        assert rel_l2(a, c) < 2e-5 or np.max(np.abs(a - c)) < 1e-6, f


@pytest.mark.parametrize('H,W,K,N', [(160, 240, 32, 1),
                                     pytest.param(384, 480, 32, 2, marks=pytest.mark.gpu),
                                     pytest.param(240, 336, 64, 1, marks=pytest.mark.gpu),
                                     pytest.param(320, 384, 128, 1, marks=pytest.mark.gpu)])
def test_joint_at_mixed_radix_sizes(backend, H, W, K, N):
    """ConvBPDNJoint (l1 + l2,1 over the three channels inside the row epilogue, re-derived from V in
    rows_fwd: sporco/admm/cbpdn.py:785-807, sporco/prox/_l21.py:51-88) at mixed-radix sizes against
    the float64 oracle; default AutoRho and a fixed rho (the emitting epilogue from the third
    iteration on); the V form bit for bit against the (Y, U) form."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(H + K)
    D = rng.randn(4, 4, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, 3, N).astype(np.float32)
    iters = 4 if backend == 'hostsim' else (8 if H * W * K * N <= 4e6 else 5)
    variants = (({}, {}), ({'AutoRho': {'Enabled': False}, 'rho': 3.0}, {'rho': 3.0, 'auto_rho': False}))
    for extra, okw in (variants[1:] if backend == 'hostsim' else variants):
        optd = dict({'MaxMainIter': iters, 'RelStopTol': 0.0}, **extra)
        b = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(optd))
        assert b._dev.uses_fused_rows() and b._fused_ok() and b._device_loop_ok()
        Y = b.solve()
        ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 3, N, 1), 0.05, mu=0.02, dtype=np.float64,
                             maxiter=iters, rel_tol=0.0, **okw)
        assert rel_l2(Y, ref['Y']) < 2e-5 and rel_l2(b.U, ref['U']) < 2e-5 and rel_l2(b.X, ref['X']) < 2e-5
        its = b.getitstat()
        for f in ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            assert rel_l2(getattr(its, f), ref[f]) < 2e-5, f
        if backend == 'hostsim':
            continue
        with env(SPORCO_AMD_NO_VFORM='1'):
            b0 = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(optd))
            Y0 = b0.solve()
        assert np.array_equal(Y, Y0)
        assert np.array_equal(np.asarray(its.ObjFun), np.asarray(b0.getitstat().ObjFun))


@pytest.mark.parametrize('H,W,K,N', [(160, 192, 3, 1),
                                     pytest.param(384, 240, 5, 2, marks=pytest.mark.gpu),
                                     pytest.param(480, 320, 63, 1, marks=pytest.mark.gpu)])
def test_gradreg_and_addmasksim_at_mixed_radix_sizes(backend, H, W, K, N):
    """ConvBPDNGradReg (the GRAD column kernel: sporco/admm/cbpdn.py:1163-1214) and AddMaskSim around
    ConvBPDN / ConvBPDNGradReg (:2287-2485; the mask bits packed one word per wave of the 16-wave row
    kernels) at mixed-radix sizes: against the generic chain, and the float64 oracle where that is
    quick."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    D, S = problem(H, W, K, N, seed=H + K + 3)
    rng = np.random.RandomState(3)
    Wm = (rng.rand(H, W, N) > 0.25).astype(np.float32)
    iters = 4
    wg = np.linspace(0.0, 2.0, K).astype(np.float32)
    small = K * N <= 8
    fields = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho')
    # --- ConvBPDNGradReg
    outs = []
    for unfused in (False, True):
        with env(**({'SPORCO_AMD_UNFUSED': '1'} if unfused else {})):
            b = cbpdn.ConvBPDNGradReg(D, S, 0.05, 0.3, cbpdn.ConvBPDNGradReg.Options(
                {'MaxMainIter': iters, 'RelStopTol': 0.0, 'GradWeight': wg}))
        assert bool(b._dev.uses_fused_rows()) == (not unfused)
        outs.append((b.solve(), b.X.copy(), b.getitstat()))
    assert rel_l2(outs[0][0], outs[1][0]) < 2e-5 and rel_l2(outs[0][1], outs[1][1]) < 2e-5
    for f in fields + ('RegGrad',):
        assert rel_l2(getattr(outs[0][2], f), getattr(outs[1][2], f)) < 1e-4, f
    if small:
        ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05, dtype=np.float64,
                             maxiter=iters, rel_tol=0.0, grad_mu=0.3, grad_weight=wg.astype(np.float64))
        assert rel_l2(outs[0][0], ref['Y']) < 1e-4
        for f in fields + ('RegGrad',):
            assert rel_l2(getattr(outs[0][2], f), ref[f]) < 1e-3, f
    # --- AddMaskSim around ConvBPDN and (GPU) around ConvBPDNGradReg
    for gradreg in ((False,) if backend == 'hostsim' else (False, True)):
        cls = cbpdn.ConvBPDNGradReg if gradreg else cbpdn.ConvBPDN
        args = (0.05, 0.3) if gradreg else (0.05,)
        optd = {'MaxMainIter': iters, 'RelStopTol': 0.0, 'NonNegCoef': True}
        runs = []
        for unfused in (False, True):
            with env(**({'SPORCO_AMD_UNFUSED': '1'} if unfused else {})):
                b = cbpdn.AddMaskSim(cls, D, S, Wm, *args, opt=cls.Options(optd))
            b.solve()
            assert bool(b.cbpdn._dev.uses_fused_rows()) == (not unfused) and b.cbpdn._fused_ok()
            runs.append(b)
        b, b0 = runs
        assert rel_l2(b.cbpdn.Y, b0.cbpdn.Y) < 2e-5 and rel_l2(b.cbpdn.U, b0.cbpdn.U) < 2e-5
        for f in fields:
            assert rel_l2(getattr(b.getitstat(), f), getattr(b0.getitstat(), f)) < 1e-4, f
        Yi = b.cbpdn.Y[..., -1]
        assert np.all(Yi[Wm.reshape(Yi.shape) != 0] == 0)
        if small and not gradreg:
            imp = np.zeros((4, 4, 1), np.float32)
            imp[0, 0] = 1
            Di = np.concatenate((D, imp), axis=2)
            ref = orc.admm_cbpdn(Di.reshape(4, 4, 1, 1, K + 1), S.reshape(H, W, 1, N, 1), 0.05, dtype=np.float64,
                                 maxiter=iters, rel_tol=0.0, nonneg=True, ams_mask=Wm.reshape(H, W, 1, N, 1))
            assert rel_l2(b.cbpdn.Y, ref['Y']) < 1e-4
            for f in fields:
                assert rel_l2(getattr(b.getitstat(), f), ref[f]) < 1e-3, f


MD_OPTS = {'default': {}, 'options': {'NonNegCoef': True, 'NoBndryCross': True, 'AuxVarObj': True, 'RelaxParam': 1.5,
                                      'AutoRho': {'Enabled': True, 'Period': 2}}}


@pytest.mark.parametrize('H,W,K,N,case', [(160, 192, 4, 1, 'default'),
                                          pytest.param(400, 240, 6, 2, 'default', marks=pytest.mark.gpu),
                                          pytest.param(400, 240, 6, 2, 'options', marks=pytest.mark.gpu),
                                          pytest.param(320, 224, 30, 1, 'default', marks=pytest.mark.gpu),
                                          # (the remaining points-per-thread counts: 18, 21 and 27, 28)
                                          pytest.param(288, 336, 8, 1, 'default', marks=pytest.mark.gpu),
                                          pytest.param(432, 448, 8, 1, 'options', marks=pytest.mark.gpu),
                                          pytest.param(384, 480, 64, 2, 'options', marks=pytest.mark.gpu)])
def test_mask_decoupling_at_mixed_radix_sizes(backend, H, W, K, N, case):
    """ConvBPDNMaskDcpl (sporco/admm/cbpdn.py:1927-2175) at mixed-radix sizes: the X-step with the
    block-0 spectrum in the signal's place and its multipliers stored (fused_cols), the block-1
    epilogue that also emits the row spectra of the new dual variable, and the dual residual's
    read-only column pass (cols_dualres) -- against the generic chain of the same library (pinned
    by the reference's fixtures, tests/test_maskdcpl.py) and, where the host finishes it, the
    float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(H + K)
    D = rng.randn(4, 4, K).astype(np.float32)
    S = rng.randn(H, W, N).astype(np.float32)
    M = (rng.rand(H, W, N) > 0.3).astype(np.float32)
    cls = cbpdn.ConvBPDNMaskDcpl
    iters = 4 if backend == 'hostsim' else 6
    optd = dict(MD_OPTS[case], MaxMainIter=iters)

    def run(generic):
        with env(**({'SPORCO_AMD_MD_GENERIC': '1'} if generic else {})):
            b = cls(D, S, 0.1, M, cls.Options(optd))
            b._dev.profile(True)
            Y1 = b.solve()
        return b, Y1, set(kernel_counts(b))

    bf, Yf, pf = run(False)
    bg, Yg, pg = run(True)
    assert {'rows_fwd', 'fused_cols_sm', 'rows_inv_post', 'setcoef_cols'} <= pf and 'sm_solve' not in pf
    assert 'sm_solve' in pg and 'fused_cols_sm' not in pg
    assert rel_l2(Yf, Yg) < 1e-4 and rel_l2(bf.X, bg.X) < 1e-4 and rel_l2(bf.U, bg.U) < 1e-4
    assert rel_l2(bf.var_y0(), bg.var_y0()) < 1e-3
    assert rel_l2(bf.reconstruct(), bg.reconstruct()) < 1e-4
    for f in TRACES:
        assert rel_l2(getattr(bf.getitstat(), f), getattr(bg.getitstat(), f)) < 1e-4, f
    if case == 'default' and H * W * K * N <= 2.5e6:
        r = orc.admm_cbpdn_maskdcpl(D.reshape(4, 4, 1, 1, K).astype(np.float64),
                                    S.reshape(H, W, 1, N, 1).astype(np.float64), 0.1,
                                    M.reshape(H, W, 1, N, 1).astype(np.float64), maxiter=iters)
        assert rel_l2(Yf, r['Y1']) < 1e-4
        for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            assert rel_l2(getattr(bf.getitstat(), f), r[f]) < 1e-4, f
