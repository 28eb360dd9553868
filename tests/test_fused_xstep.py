"""The fused X-step (csc_fused.hip: column FFT + Sherman-Morrison + column IFFT in
registers, tile-major intermediates) against (a) the NumPy oracle and (b) the
unfused kernel chain of the same library.

The fused path engages for float32, H in {128, 256, 512}, even K <= 64; the widths
are kept tiny so that the CPU fiber simulator finishes in seconds (the row
kernels are size-generic, so W does not matter to the code under test).

Tolerances: 1e-4 relative l2 on Y against the float64 oracle (the BASELINE bar);
fused vs unfused float32 runs agree to 2e-5 after 12 adaptive-rho iterations
(same arithmetic up to summation order and a 1-ulp reciprocal).
"""

import os

import numpy as np
import pytest

from conftest import rel_l2


def problem(H, W, K, N, seed, C=None):
    rng = np.random.RandomState(seed)
    D = rng.randn(4, 4, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    shape = (H, W, N) if C is None else (H, W, C, N)
    S = rng.randn(*shape).astype(np.float32)
    return D, S


def solve(D, S, optd, unfused=False, joint=False):
    from sporco_amd.admm import cbpdn
    if unfused:
        os.environ['SPORCO_AMD_UNFUSED'] = '1'
    try:
        if joint:
            b = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(optd))
        else:
            b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    finally:
        os.environ.pop('SPORCO_AMD_UNFUSED', None)
    Y = b.solve()
    return b, Y


@pytest.mark.parametrize('H,W,K,N', [(256, 16, 8, 2), (512, 12, 64, 1), (512, 8, 6, 3),
                                     pytest.param(256, 10, 64, 2, marks=pytest.mark.gpu)])
def test_fused_matches_oracle_and_unfused(backend, H, W, K, N):
    from oracle import cbpdn_oracle as orc
    D, S = problem(H, W, K, N, seed=H + K)
    optd = {'MaxMainIter': 12, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd)
    b0, Y0 = solve(D, S, optd, unfused=True)
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05,
                         dtype=np.float64, maxiter=12, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-4
    assert rel_l2(Y, Y0) < 2e-5
    its, its0 = b.getitstat(), b0.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-3, f
        assert rel_l2(getattr(its, f), getattr(its0, f)) < 1e-4, f
    # X and Xf stay available after a fused step (Xf is rebuilt from X on demand)
    assert rel_l2(b.X, ref['X']) < 1e-4
    assert rel_l2(b.Xf, np.fft.rfftn(b.X.astype(np.float64), axes=(0, 1))) < 1e-5
    assert rel_l2(b.reconstruct(), b0.reconstruct()) < 2e-5


def test_fused_joint_multichannel(backend):
    from oracle import cbpdn_oracle as orc
    H, W, K, N, C = 256, 8, 16, 2, 3
    D, S = problem(H, W, K, N, seed=5, C=C)
    optd = {'MaxMainIter': 10, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd, joint=True)
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, C, N, 1), 0.05, mu=0.02,
                         dtype=np.float64, maxiter=10, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-4
    assert rel_l2(b.getitstat().ObjFun, ref['ObjFun']) < 1e-3


def test_joint_l21_inside_the_row_epilogue(backend):
    """ConvBPDNJoint on the three-launch path: the l2 norm over the channels that prox_sl1l2
    needs (cbpdn.py:785-794, prox/_l21.py:51-88) is taken inside `rows_inv_post` -- lanes =
    (channel, filter pair), channel sums by permlane swaps -- so no separate epilogue pass
    runs.  Against the float64 oracle; a weight array falls back to the separate epilogue."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    H, W, C, K = (128, 256, 3, 32) if backend == 'hostsim' else (256, 256, 3, 32)
    N, iters = (1, 2) if backend == 'hostsim' else (3, 8)
    D, S = problem(H, W, K, N, seed=31, C=C)
    opt = cbpdn.ConvBPDNJoint.Options({'MaxMainIter': iters, 'RelStopTol': 0.0})
    b = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, opt)
    assert b._dev.uses_fused_rows() and b._fused_ok()
    b.profile(True)
    os.environ['SPORCO_AMD_HOST_LOOP'] = '1'       # (so that the counters name what ran)
    try:
        Y = b.solve()
    finally:
        os.environ.pop('SPORCO_AMD_HOST_LOOP', None)
    cnt = {k: v[1] for k, v in b.profile_read().items() if v[1] > 0}
    assert sum(cnt.get(k, 0) for k in ('rows_inv_post', 'rows_inv_post_emit', 'rows_inv_post_v',
                                       'rows_inv_post_v_emit')) == iters
    assert 'admm_post' not in cnt and 'fft_c2r_rows' not in cnt
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, C, N, 1), 0.05, mu=0.02,
                         dtype=np.float64, maxiter=iters, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-5 and rel_l2(b.U, ref['U']) < 1e-5
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-5, f
    if backend == 'hostsim':
        return
    # device-driven loop == host loop, bit for bit; NonNegCoef + gEvalY through the same kernel
    for extra in ({}, {'NonNegCoef': True, 'AuxVarObj': False, 'gEvalY': True}):
        optd = dict({'MaxMainIter': iters, 'RelStopTol': 0.0}, **extra)
        bd = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(optd))
        assert bd._device_loop_ok()
        Yd = bd.solve()
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
        try:
            bh = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(optd))
            Yh = bh.solve()
        finally:
            os.environ.pop('SPORCO_AMD_HOST_LOOP', None)
        assert np.array_equal(Yd, Yh)
        assert np.array_equal(np.asarray(bd.getitstat().RegL21), np.asarray(bh.getitstat().RegL21))
        refx = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, C, N, 1), 0.05, mu=0.02,
                              dtype=np.float64, maxiter=iters, rel_tol=0.0,
                              nonneg=bool(extra), gevaly=bool(extra))
        assert rel_l2(Yd, refx['Y']) < 1e-5
        assert rel_l2(bd.getitstat().RegL21, refx['RegL21']) < 1e-5
    # an L21Weight array: the separate register-resident epilogue, same answer
    w21 = (0.5 + np.random.RandomState(2).rand(1, 1, N, K)).astype(np.float32)
    bw = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(
        {'MaxMainIter': 3, 'RelStopTol': 0.0, 'L21Weight': w21}))
    bw.profile(True)
    Yw = bw.solve()
    assert {k: v[1] for k, v in bw.profile_read().items() if v[1] > 0}.get('admm_post', 0) == 3
    refw = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, C, N, 1), 0.05, mu=0.02,
                          dtype=np.float64, maxiter=3, rel_tol=0.0, wl21=w21[:, :, np.newaxis])
    assert rel_l2(Yw, refw['Y']) < 1e-5


def test_fused_fixed_rho_fastsolve_and_setdict(backend):
    """FastSolve (no sums read back) and a dictionary change between solves."""
    H, W, K, N = 256, 8, 8, 2
    D, S = problem(H, W, K, N, seed=9)
    optd = {'MaxMainIter': 6, 'RelStopTol': 0.0, 'rho': 2.0, 'FastSolve': True,
            'AutoRho': {'Enabled': False}}
    b, Y = solve(D, S, optd)
    b0, Y0 = solve(D, S, optd, unfused=True)
    assert rel_l2(Y, Y0) < 1e-5
    D2 = D[..., ::-1].copy()
    for s in (b, b0):
        s.setdict(D2.reshape(s.cri.shpD))
        s.solve()
    assert rel_l2(b.Y, b0.Y) < 1e-5


# ---------------------------------------------------------------------------
# the three-launch iteration (csc_rows.hip + csc_fused.hip): H and W in {256, 512}
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('H,W,K,N', [(256, 256, 4, 1), (256, 512, 6, 1),
                                     # 128 in either direction (32 x 4 splits, round 2)
                                     (128, 128, 4, 1), (128, 256, 6, 2), (256, 128, 4, 1),
                                     pytest.param(128, 512, 64, 2, marks=pytest.mark.gpu),
                                     pytest.param(512, 128, 64, 2, marks=pytest.mark.gpu),
                                     pytest.param(128, 128, 64, 8, marks=pytest.mark.gpu)])
def test_three_launch_iteration_matches_oracle(backend, H, W, K, N):
    from oracle import cbpdn_oracle as orc
    D, S = problem(H, W, K, N, seed=H + W + K)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd)
    assert b._dev.uses_fused_rows()
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05,
                         dtype=np.float64, maxiter=3, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-5
    assert rel_l2(b.U, ref['U']) < 1e-5
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-5, f
    # X never left the registers during the iterations: it is rebuilt on demand from
    # the previous iterate, and Xf from X
    assert rel_l2(b.X, ref['X']) < 1e-5
    assert rel_l2(b.Xf, np.fft.rfftn(ref['X'], axes=(0, 1))) < 1e-5
    if backend == 'hostsim' and W == 512:
        return       # (keeps the CPU suite short)
    # the solver keeps going from where it stopped (admm.py:331)
    b.solve()
    ref8 = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05,
                          dtype=np.float64, maxiter=6, rel_tol=0.0)
    assert rel_l2(b.Y, ref8['Y']) < 2e-5


def test_three_launch_weights_nonneg_nobndry(backend):
    """L1Weight array + NonNegCoef + NoBndryCross through the GENERAL epilogue."""
    H, W, K, N = 256, 256, 4, 2
    D, S = problem(H, W, K, N, seed=21)
    rng = np.random.RandomState(3)
    wl1 = (0.5 + rng.rand(H, W, 1, 1, K)).astype(np.float32)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0, 'NonNegCoef': True, 'NoBndryCross': True,
            'L1Weight': wl1}
    b, Y = solve(D, S, optd)
    b0, Y0 = solve(D, S, optd, unfused=True)
    assert b._dev.uses_fused_rows() and not b0._dev.uses_fused_rows()
    assert rel_l2(Y, Y0) < 1e-5
    assert np.all(Y >= 0) and np.all(Y[-3:] == 0) and np.all(Y[:, -3:] == 0)
    for f in ('ObjFun', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(b.getitstat(), f), getattr(b0.getitstat(), f)) < 1e-5, f


def test_size_128_options_joint_gradreg_and_many_iterations(backend):
    """H = W = 128 through the option-dependent kernel variants: the GENERAL epilogue (L1Weight
    array, NonNegCoef, NoBndryCross), the l2,1 epilogue (ConvBPDNJoint), the gradient-regularised
    column kernel, and 14 / 30 default-option iterations (rho changes, then the emitting epilogue)
    against the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    H, W, K, N = 128, 128, 4, 2
    D, S = problem(H, W, K, N, seed=5)
    rng = np.random.RandomState(3)
    wl1 = (0.5 + rng.rand(H, W, 1, 1, K)).astype(np.float32)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0, 'NonNegCoef': True, 'NoBndryCross': True,
            'L1Weight': wl1}
    b, Y = solve(D, S, optd)
    b0, Y0 = solve(D, S, optd, unfused=True)
    assert b._dev.uses_fused_rows() and not b0._dev.uses_fused_rows()
    assert rel_l2(Y, Y0) < 1e-5 and np.all(Y >= 0) and np.all(Y[-3:] == 0) and np.all(Y[:, -3:] == 0)
    # default options long enough for rho to settle and the emitting epilogue to run
    many = 14 if backend == 'hostsim' else 30
    optd = {'MaxMainIter': many, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd)
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05,
                         dtype=np.float64, maxiter=many, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-4
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(b.getitstat(), f), ref[f]) < 1e-3, f
    # ConvBPDNJoint, C = 3, K = 32 (the joint epilogue needs K % 32 == 0)
    Dj, Sj = problem(H, W, 32, 1, seed=7, C=3)
    bj, Yj = solve(Dj, Sj, {'MaxMainIter': 3, 'RelStopTol': 0.0}, joint=True)
    refj = orc.admm_cbpdn(Dj.reshape(4, 4, 1, 1, 32), Sj.reshape(H, W, 3, 1, 1), 0.05, mu=0.02,
                          dtype=np.float64, maxiter=3, rel_tol=0.0)
    assert bj._dev.uses_fused_rows() and rel_l2(Yj, refj['Y']) < 1e-5
    # ConvBPDNGradReg
    bg = cbpdn.ConvBPDNGradReg(D, S, 0.05, 0.3, cbpdn.ConvBPDNGradReg.Options(
        {'MaxMainIter': 3, 'RelStopTol': 0.0}))
    Yg = bg.solve()
    refg = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05, grad_mu=0.3,
                          dtype=np.float64, maxiter=3, rel_tol=0.0)
    assert bg._dev.uses_fused_rows() and rel_l2(Yg, refg['Y']) < 1e-5


def test_no_x_hint_and_pickle(backend):
    if backend == 'hostsim':
        pytest.skip("kept for the GPU run: the CPU suite covers these kernels in other cases")
    import pickle
    from sporco_amd import _lib
    H, W, K, N = 256, 256, 4, 1
    D, S = problem(H, W, K, N, seed=33)
    optd = {'MaxMainIter': 2, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd)
    b2 = pickle.loads(pickle.dumps(b))
    assert np.array_equal(b2.X, b.X) and np.array_equal(b2.Y, b.Y)
    b.solve()
    b2.solve()
    assert np.array_equal(b.Y, b2.Y)          # pickle round trip continues bit-identically
    b._no_x = True                            # DictLearn's promise: X is not read
    b.solve()
    with pytest.raises(_lib.BackendError):
        b.X
    assert np.all(np.isfinite(b.Y)) and np.any(b.Y != 0)     # Y stays available


def test_speculative_rows_fwd_is_bit_identical(backend):
    """With rho unchanged, rows_inv_post also emits the next iteration's rows_fwd output
    (two passes fewer per iteration); the iterates must not change by a single bit."""
    H, W, K, N = 256, 256, 4, 1
    D, S = problem(H, W, K, N, seed=91)
    optd = {'MaxMainIter': (3 if backend == 'hostsim' else 6), 'RelStopTol': 0.0, 'rho': 1.5,
            'AutoRho': {'Enabled': False}}
    b, Y = solve(D, S, optd)
    os.environ['SPORCO_AMD_NO_SPECULATION'] = '1'
    try:
        b0, Y0 = solve(D, S, optd)
    finally:
        os.environ.pop('SPORCO_AMD_NO_SPECULATION', None)
    assert np.array_equal(Y, Y0) and np.array_equal(b.U, b0.U) and np.array_equal(b.X, b0.X)
    assert np.array_equal(np.asarray(b.getitstat().ObjFun), np.asarray(b0.getitstat().ObjFun))
    if backend == 'hostsim':
        return       # (keeps the CPU suite short; the random-walk test covers this on the GPU)
    # a dictionary change invalidates the speculated spectra
    D2 = D[..., ::-1].copy()
    for s in (b, b0):
        s.setdict(D2.reshape(s.cri.shpD))
        s.solve()
    assert np.array_equal(b.Y, b0.Y)


def test_joint_at_fused_row_sizes(backend):
    """ConvBPDNJoint where the row kernels engage too (X-step through rows_fwd / fused
    columns / row inverse, l2,1 epilogue generic)."""
    if backend == 'hostsim':
        pytest.skip("kept for the GPU run: the CPU suite covers these kernels in other cases")
    from oracle import cbpdn_oracle as orc
    H, W, K, N, C = 256, 256, 4, 1, 3
    D, S = problem(H, W, K, N, seed=15, C=C)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd, joint=True)
    assert b._dev.uses_fused_rows()
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, C, N, 1), 0.05, mu=0.02,
                         dtype=np.float64, maxiter=3, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-5
    assert rel_l2(b.X, ref['X']) < 1e-5
    assert rel_l2(b.getitstat().ObjFun, ref['ObjFun']) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_state_machine_random_walk(gpu_backend, seed):
    """Random sequences of solves, reads, writes and dictionary changes drive the fused
    path's lazy state (ping-pong iterates, on-demand X / Xf, speculated row spectra)
    through its transitions; a solver on the generic chain must stay in step."""
    import pickle
    H, W, K, N = 256, 256, 6, 2
    D, S = problem(H, W, K, N, seed=100 + seed)
    rng = np.random.RandomState(seed)
    fixed = seed == 1
    optd = {'MaxMainIter': 2, 'RelStopTol': 0.0}
    if fixed:
        optd.update({'rho': 2.0, 'AutoRho': {'Enabled': False}})
    b, _ = solve(D, S, optd)
    g, _ = solve(D, S, optd, unfused=True)
    assert b._dev.uses_fused_rows() and not g._dev.uses_fused_cols()
    tol = 2e-5
    for step in range(14):
        op = rng.randint(8)
        if op == 0:
            n = int(rng.randint(1, 4))
            for s in (b, g):
                s.opt['MaxMainIter'] = n
                s.solve()
        elif op == 1:
            assert rel_l2(b.X, g.X) < tol, (step, 'X')
        elif op == 2:
            assert rel_l2(b.Xf, g.Xf) < tol, (step, 'Xf')
        elif op == 3:
            D2 = (D * (1.0 + 0.1 * rng.randn(1, 1, K))).astype(np.float32)
            for s in (b, g):
                s.setdict(D2.reshape(s.cri.shpD))
        elif op == 4:
            Ynew = (b.Y * np.float32(0.9)).copy()
            for s in (b, g):
                s.Y = Ynew
        elif op == 5:
            assert rel_l2(b.reconstruct(), g.reconstruct()) < tol, (step, 'recon')
        elif op == 6:
            b = pickle.loads(pickle.dumps(b))
        else:
            assert rel_l2(b.U, g.U) < tol, (step, 'U')
        assert rel_l2(b.Y, g.Y) < tol, (step, op, 'Y')
    for s in (b, g):
        s.opt['MaxMainIter'] = 3
        s.solve()
    assert rel_l2(b.Y, g.Y) < tol and rel_l2(b.X, g.X) < tol
    its, itg = b.getitstat(), g.getitstat()
    assert len(its.ObjFun) == len(itg.ObjFun)
    assert rel_l2(its.ObjFun, itg.ObjFun) < 1e-4 and rel_l2(its.Rho, itg.Rho) < 1e-4


# ---------------------------------------------------------------------------
# K = 64 * NH filters: the column pass runs as two kernels over 64-filter slabs
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('H,W,K,N,C', [(256, 256, 128, 1, None),
                                       (128, 128, 128, 1, None),
                                       # 64 < K <= 72: one column kernel on the first 64
                                       # filters, the tail through the generic column FFT
                                       (256, 256, 70, 1, None),
                                       pytest.param(512, 512, 66, 2, None, marks=pytest.mark.gpu),
                                       pytest.param(256, 256, 72, 1, 3, marks=pytest.mark.gpu),
                                       pytest.param(512, 512, 128, 1, None, marks=pytest.mark.gpu),
                                       pytest.param(256, 512, 192, 1, None, marks=pytest.mark.gpu),
                                       pytest.param(256, 256, 96, 2, None, marks=pytest.mark.gpu),
                                       # four cooperating slab workgroups per tile
                                       pytest.param(256, 256, 256, 1, None, marks=pytest.mark.gpu),
                                       pytest.param(512, 256, 250, 1, None, marks=pytest.mark.gpu),
                                       pytest.param(256, 256, 128, 1, 3, marks=pytest.mark.gpu)])
def test_slab_column_pass_many_filters(backend, H, W, K, N, C):
    from oracle import cbpdn_oracle as orc
    D, S = problem(H, W, K, N, seed=H + K, C=C)
    iters = 2 if backend == 'hostsim' else 3
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd, joint=C is not None)
    # (K > 72: slab kernels in both solvers; 64 < K <= 72: the ADMM tail form, FISTA staged)
    assert b._dev.uses_fused_rows() and bool(b._dev.uses_fused_pgm()) == (K > 72)
    if H * W * K * N * (C or 1) > 2 ** 25:
        # (not reached by the cases above: every one of them, including ConvBPDNJoint with
        # K = 128, C = 3 -- 25 M elements -- is checked against the float64 oracle)
        b0, Y0 = solve(D, S, optd, unfused=True, joint=C is not None)
        ref = {'Y': Y0, 'X': b0.X}
        ref.update({f: getattr(b0.getitstat(), f)
                    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho')})
    else:
        ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, C or 1, N, 1), 0.05,
                             mu=(0.02 if C else None), dtype=np.float64, maxiter=iters,
                             rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-5
    assert rel_l2(b.X, ref['X']) < 1e-5
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-5, f


# ---------------------------------------------------------------------------
# ConvBPDNGradReg on the fused kernels (GRAD variant of the column kernel)
# ---------------------------------------------------------------------------
def solve_gradreg(D, S, optd, mu=0.3, unfused=False):
    from sporco_amd.admm import cbpdn
    if unfused:
        os.environ['SPORCO_AMD_UNFUSED'] = '1'
    try:
        b = cbpdn.ConvBPDNGradReg(D, S, 0.05, mu, cbpdn.ConvBPDNGradReg.Options(optd))
    finally:
        os.environ.pop('SPORCO_AMD_UNFUSED', None)
    return b, b.solve()


@pytest.mark.parametrize('H,W,K,N,weights', [
    (256, 12, 8, 2, False), (512, 8, 6, 1, True),
    # 64 < K <= 72: the gradient-regularised column kernel on the first 64 filters + tail
    pytest.param(256, 256, 66, 1, True, marks=pytest.mark.gpu),
    pytest.param(512, 512, 70, 2, True, marks=pytest.mark.gpu),
    # more filters: the gradient-regularised slab kernels
    pytest.param(256, 256, 128, 1, True, marks=pytest.mark.gpu),
    pytest.param(512, 256, 96, 2, False, marks=pytest.mark.gpu),
    pytest.param(256, 256, 4, 1, True, marks=pytest.mark.gpu),
    pytest.param(512, 512, 64, 2, True, marks=pytest.mark.gpu)])
def test_fused_gradreg_matches_oracle_and_unfused(backend, H, W, K, N, weights):
    from oracle import cbpdn_oracle as orc
    D, S = problem(H, W, K, N, seed=H + K + 1)
    iters = 4 if W >= 256 else 10
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0}
    wg = None
    if weights:
        wg = np.linspace(0.0, 2.0, K).astype(np.float32)
        optd['GradWeight'] = wg
    b, Y = solve_gradreg(D, S, optd)
    assert b._dev.uses_fused_cols()
    b0, Y0 = solve_gradreg(D, S, optd, unfused=True)
    assert rel_l2(Y, Y0) < 2e-5
    its, its0 = b.getitstat(), b0.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'RegGrad', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), getattr(its0, f)) < 1e-4, f
    assert rel_l2(b.X, b0.X) < 2e-5
    if W * K * N > 512 * 64:
        return           # (the float64 oracle would take minutes at this size)
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05,
                         dtype=np.float64, maxiter=iters, rel_tol=0.0, grad_mu=0.3,
                         grad_weight=1.0 if wg is None else wg.astype(np.float64))
    assert rel_l2(Y, ref['Y']) < 1e-4
    assert rel_l2(b.X, ref['X']) < 1e-4
    for f in ('ObjFun', 'DFid', 'RegL1', 'RegGrad', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-3, f


# ---------------------------------------------------------------------------
# AddMaskSim on the three-launch iteration (mask handling inside rows_inv_post)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('H,W,K,N,gradreg', [
    pytest.param(256, 256, 3, 2, False, marks=pytest.mark.gpu),
    pytest.param(256, 256, 5, 1, True, marks=pytest.mark.gpu),
    pytest.param(512, 512, 63, 3, False, marks=pytest.mark.gpu),
    # a 64-filter dictionary: 65 filters, padded to 66, column pass in tail mode, the impulse
    # in the first half of the last filter pair
    pytest.param(256, 256, 64, 2, False, marks=pytest.mark.gpu),
    pytest.param(512, 256, 64, 1, True, marks=pytest.mark.gpu)])
def test_fused_ams_matches_oracle_and_unfused(backend, H, W, K, N, gradreg):
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    D, S = problem(H, W, K, N, seed=H + K + 7)
    rng = np.random.RandomState(3)
    Wm = (rng.rand(H, W, N) > 0.25).astype(np.float32)
    iters = 4
    cls = cbpdn.ConvBPDNGradReg if gradreg else cbpdn.ConvBPDN
    args = (0.05, 0.3) if gradreg else (0.05,)
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0, 'NonNegCoef': True}

    def run(unfused):
        if unfused:
            os.environ['SPORCO_AMD_UNFUSED'] = '1'
        try:
            b = cbpdn.AddMaskSim(cls, D, S, Wm, *args, opt=cls.Options(optd))
        finally:
            os.environ.pop('SPORCO_AMD_UNFUSED', None)
        b.solve()
        return b

    b, b0 = run(False), run(True)
    assert b.cbpdn._dev.uses_fused_rows() and not b0.cbpdn._dev.uses_fused_rows()
    assert rel_l2(b.cbpdn.Y, b0.cbpdn.Y) < 2e-5
    assert rel_l2(b.cbpdn.U, b0.cbpdn.U) < 2e-5
    its, its0 = b.getitstat(), b0.getitstat()
    fields = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho')
    for f in fields:
        assert rel_l2(getattr(its, f), getattr(its0, f)) < 1e-4, f
    # the impulse slice is zero exactly where the mask is set, and only there shrunk
    Yi = b.cbpdn.Y[..., -1]
    assert np.all(Yi[Wm.reshape(Yi.shape) != 0] == 0)
    if K * N > 16:
        return           # (the float64 oracle would take minutes at this size)
    imp = np.zeros((4, 4, 1), np.float32)
    imp[0, 0] = 1
    Di = np.concatenate((D, imp), axis=2)
    kw = dict(grad_mu=0.3) if gradreg else {}
    ref = orc.admm_cbpdn(Di.reshape(4, 4, 1, 1, K + 1), S.reshape(H, W, 1, N, 1), 0.05,
                         dtype=np.float64, maxiter=iters, rel_tol=0.0, nonneg=True,
                         ams_mask=Wm.reshape(H, W, 1, N, 1), **kw)
    assert rel_l2(b.cbpdn.Y, ref['Y']) < 1e-4
    for f in fields:
        assert rel_l2(getattr(its, f), ref[f]) < 1e-3, f


# ---------------------------------------------------------------------------
# odd filter counts: the handle pads the filter axis with one all-zero filter to reach the
# fused kernels; host arrays keep the caller's K
# ---------------------------------------------------------------------------
def test_odd_filter_count_is_padded_on_device(backend):
    """AddMaskSim(ConvBPDNGradReg) with K = 4 (+ impulse = 5 filters, 6 on the device),
    per-filter L1Weight and GradWeight: everything that carries a filter axis across the
    ABI, against the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    H, W, K, N = 256, 256, 4, 1
    D, S = problem(H, W, K, N, seed=77)
    rng = np.random.RandomState(5)
    Wm = (rng.rand(H, W) > 0.25).astype(np.float32)
    wl1 = np.linspace(0.5, 1.5, K + 1).astype(np.float32).reshape(1, 1, 1, 1, K + 1)
    wg = np.array([0.0, 1.0, 0.5, 2.0, 1.0], np.float32)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0, 'L1Weight': wl1, 'GradWeight': wg}
    b = cbpdn.AddMaskSim(cbpdn.ConvBPDNGradReg, D, S, Wm, 0.05, 0.3,
                         opt=cbpdn.ConvBPDNGradReg.Options(optd))
    c = b.cbpdn
    assert c._dev.query(_lib.QUERY_DEVICE_FILTERS) == K + 2 and c._dev.uses_fused_rows()
    b.solve()
    assert c.Y.shape == (H, W, 1, N, K + 1) and c.Xf.shape[-1] == K + 1
    imp = np.zeros((4, 4, 1), np.float32)
    imp[0, 0] = 1
    Di = np.concatenate((D, imp), axis=2)
    ref = orc.admm_cbpdn(Di.reshape(4, 4, 1, 1, K + 1), S.reshape(H, W, 1, N, 1), 0.05,
                         dtype=np.float64, maxiter=3, rel_tol=0.0, grad_mu=0.3,
                         grad_weight=wg.astype(np.float64), wl1=wl1.astype(np.float64),
                         ams_mask=Wm.reshape(H, W, 1, 1, 1))
    for key in ('Y', 'U', 'X'):
        assert rel_l2(getattr(c, key), ref[key]) < 1e-5, key
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'RegGrad', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-5, f
    # state written back through the ABI lands in the right filters
    Y = c.Y.copy()
    c.Y = Y
    assert np.array_equal(c.Y, Y)


@pytest.mark.gpu
@pytest.mark.parametrize('K', [7, 63, 65])
def test_odd_filter_count_matches_unpadded(backend, K):
    H, W, N = 256, 256, 2
    D, S = problem(H, W, K, N, seed=K)
    optd = {'MaxMainIter': 8, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd)
    assert b._dev.uses_fused_rows()
    os.environ['SPORCO_AMD_NO_PAD'] = '1'
    try:
        b0, Y0 = solve(D, S, optd)
    finally:
        os.environ.pop('SPORCO_AMD_NO_PAD', None)
    assert not b0._dev.uses_fused_rows()
    assert Y.shape == Y0.shape == (H, W, 1, N, K)
    assert rel_l2(Y, Y0) < 2e-5
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(b.getitstat(), f), getattr(b0.getitstat(), f)) < 1e-4, f
    assert rel_l2(b.X, b0.X) < 2e-5 and rel_l2(b.Xf, b0.Xf) < 2e-5
    assert rel_l2(b.reconstruct(), b0.reconstruct()) < 2e-5


# ---------------------------------------------------------------------------
# multi-channel dictionaries on the three-launch iteration (csc_fused_mc.hip)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('H,W,C,K,N', [pytest.param(256, 256, 3, 4, 1, marks=pytest.mark.gpu),
                                       (128, 128, 3, 4, 1),
                                       pytest.param(256, 256, 2, 6, 2, marks=pytest.mark.gpu),
                                       pytest.param(512, 256, 4, 8, 1, marks=pytest.mark.gpu),
                                       pytest.param(512, 512, 3, 64, 3, marks=pytest.mark.gpu)])
def test_fused_multichannel_dictionary(backend, H, W, C, K, N):
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(H + C + K)
    D = rng.randn(4, 4, C, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1, 2), keepdims=True))
    S = rng.randn(H, W, C, N).astype(np.float32)
    iters = 2 if backend == 'hostsim' else 3
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0}
    kw = {}
    if backend == 'hostsim':
        # (one rho: the wave-level reductions of the B-matrix kernel are slow to simulate)
        optd.update({'rho': 4.0, 'AutoRho': {'Enabled': False}})
        kw = dict(rho=4.0, auto_rho=False)

    def run(unfused):
        if unfused:
            os.environ['SPORCO_AMD_UNFUSED'] = '1'
        try:
            b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
        finally:
            os.environ.pop('SPORCO_AMD_UNFUSED', None)
        b.solve()
        return b

    b = run(False)
    assert b._dev.uses_fused_rows()
    fields = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho')
    if K * N * C <= 16:
        ref = orc.admm_cbpdn(D.reshape(4, 4, C, 1, K), S.reshape(H, W, C, N, 1), 0.05,
                             dtype=np.float64, maxiter=iters, rel_tol=0.0, **kw)
        assert rel_l2(b.Y, ref['Y']) < 1e-5 and rel_l2(b.U, ref['U']) < 1e-5
        assert rel_l2(b.X, ref['X']) < 1e-5
        for f in fields:
            assert rel_l2(getattr(b.getitstat(), f), ref[f]) < 1e-5, f
    if backend == 'hostsim':
        return
    b0 = run(True)
    assert not b0._dev.uses_fused_rows()
    assert rel_l2(b.Y, b0.Y) < 1e-5 and rel_l2(b.X, b0.X) < 1e-5
    assert rel_l2(b.Xf, b0.Xf) < 1e-5
    assert rel_l2(b.reconstruct(), b0.reconstruct()) < 1e-5
    for f in fields:
        assert rel_l2(getattr(b.getitstat(), f), getattr(b0.getitstat(), f)) < 1e-5, f
