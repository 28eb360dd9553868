"""Oracle / reference parity of the kernels the benchmarks run, at the benchmark shapes.

Every BASELINE.json configuration is timed on a particular set of kernel instantiations
(512-point register FFTs with 64 filters per wave row, the 64-filter FISTA kernels, the
64-filter tile-major dictionary gradient, the two-slab column pass of K = 128).  These tests
put the float64 oracle (and fixtures written by the unmodified reference,
oracle/make_golden.py `config2` / `tol`) on exactly those instantiations:

* config 2  ConvBPDN 512x512, K = 64: N = 2 with default options (both the plain and the
  spectrum-emitting variant of `rows_inv_post`), and the full N = 32 batch whose first and
  last image are compared with the oracle run on those two images alone (fixed rho, so the
  images are independent);
* config 4  fused FISTA 512x512, K = 64, N = 2;
* config 5  tile-major D-step 256x256, K = 64, N = 4 and a few outer iterations of
  ConvBPDNDictLearn at K = 64;
* config 3  ConvBPDNJoint, K = 128, C = 3 (slab column pass + joint epilogue), 256x256, N = 1;
* the stopping iteration of the fused float32 path on the sparse-synthesis input of the
  time-to-tolerance measurement, against the reference's own count.

Tolerances: coefficient maps <= 1e-4 relative l2 against float64 (BASELINE.json), traces
<= 1e-3 (observed: see DESIGN.md section 2).
"""

import os
import sys

import numpy as np
import pytest

from conftest import REPO, load_golden, rel_l2

pytestmark = pytest.mark.gpu

if REPO not in sys.path:
    sys.path.insert(0, REPO)

TRACES = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')


def kernel_counts(b):
    return {k: v[1] for k, v in b.profile_read().items() if v[1] > 0}


class host_loop(object):
    """Run solve() with the per-iteration host loop (which launches exactly the kernels that
    execute, so that the profile counters say which variants ran); the default device-driven
    loop enqueues both epilogue variants every iteration and lets the device pick."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        if self.on:
            os.environ['SPORCO_AMD_HOST_LOOP'] = '1'

    def __exit__(self, *exc):
        os.environ.pop('SPORCO_AMD_HOST_LOOP', None)


_oracle_cache = {}


def test_config2_three_launch_kernels_vs_reference_fixture(gpu_backend):
    """ConvBPDN 512x512, K=64, N=2, float32, default options, 10 iterations, against the
    traces and iterate the reference itself produced for this input in float64 and float32
    (tests/golden/admm_config2_n2_*.npz)."""
    import bench
    from sporco_amd.admm import cbpdn
    D, S = bench.make_problem(512, 512, 64, 2, 0)
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 10, 'RelStopTol': 0.0})
    b = cbpdn.ConvBPDN(D, S, 0.05, opt)
    assert b._dev.uses_fused_rows() and b._fused_ok() and b._device_loop_ok()
    b.profile(True)
    Y = b.solve()
    cnt = kernel_counts(b)
    assert cnt.get('rows_fwd', 0) > 0 and cnt.get('fused_cols_sm', 0) == 10
    assert not any(k in cnt for k in ('fft_r2c_rows', 'sm_solve', 'admm_post'))
    its = b.getitstat()
    g64, g32 = load_golden('admm_config2_n2_f64'), load_golden('admm_config2_n2_f32')
    assert rel_l2(Y[::16, ::16], g64['Y_sub']) < 1e-4
    assert abs(np.linalg.norm(Y.astype(np.float64)) - float(g64['Y_l2'])) < 1e-4 * float(g64['Y_l2'])
    for f in TRACES:
        assert rel_l2(getattr(its, f), g64['it_' + f]) < 1e-3, f
        # the float32 reference run itself is no closer to float64 than we are (x10 margin)
        assert rel_l2(getattr(its, f), g64['it_' + f]) < \
            10 * max(rel_l2(g32['it_' + f], g64['it_' + f]), 1e-6), f


@pytest.mark.parametrize('loop', ['device', 'host'])
@pytest.mark.parametrize('period', [1, 4])
def test_config2_three_launch_kernels_vs_oracle(gpu_backend, period, loop):
    """Same shape against the float64 oracle on the full arrays (every element of Y, U, X).
    AutoRho period 4 leaves rho alone between updates, so the iteration runs the
    spectrum-emitting epilogue (`rows_inv_post<..., EMIT_T=true>`) and skips `rows_fwd`;
    period 1 (the default) runs the plain variant while rho moves."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(202 + period)
    D = rng.randn(8, 8, 64).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(512, 512, 2).astype(np.float32)
    iters = 8 if period == 1 else 10
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': iters, 'RelStopTol': 0.0,
                                  'AutoRho': {'Period': period}})
    b = cbpdn.ConvBPDN(D, S, 0.05, opt)
    assert b._dev.uses_fused_rows() and b._fused_ok()
    b.profile(True)
    with host_loop(loop == 'host'):
        Y = b.solve()
    cnt = kernel_counts(b)
    # (the kernels of the single-array state, csc_rows.h, count with their (Y, U) twins)
    for name in ('rows_fwd', 'rows_inv_post', 'rows_inv_post_emit'):
        twin = name.replace('rows_fwd', 'rows_fwd_v').replace('rows_inv_post', 'rows_inv_post_v')
        cnt[name] = cnt.get(name, 0) + cnt.get(twin, 0)
    if loop == 'device':
        assert cnt.get('fused_cols_sm', 0) == iters
    elif period == 1:
        assert cnt.get('rows_inv_post', 0) >= 6
    else:
        assert cnt.get('rows_inv_post_emit', 0) >= 4 and cnt.get('rows_inv_post', 0) >= 2
        assert cnt.get('rows_fwd', 0) < iters          # some forward passes were skipped
    if period not in _oracle_cache:
        _oracle_cache[period] = orc.admm_cbpdn(
            D.reshape(8, 8, 1, 1, 64), S.reshape(512, 512, 1, 2, 1), 0.05, dtype=np.float64,
            maxiter=iters, rel_tol=0.0, rho_period=period)
    ref = _oracle_cache[period]
    assert rel_l2(Y, ref['Y']) < 1e-4
    assert rel_l2(b.U, ref['U']) < 1e-4
    assert rel_l2(b.X, ref['X']) < 1e-4
    its = b.getitstat()
    for f in TRACES:
        assert rel_l2(getattr(its, f), ref[f]) < 1e-3, f


def test_config2_full_batch_first_and_last_image_vs_oracle(gpu_backend):
    """The benchmarked problem itself (512x512, K=64, N=32: bench.make_problem) at fixed rho
    for 6 iterations; images 0 and 31 of the result against the oracle run on those two
    images (the iteration is independent per image when rho is fixed)."""
    import bench
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    D, S = bench.make_problem(512, 512, 64, 32, 0)
    optd = {'MaxMainIter': 6, 'RelStopTol': 0.0, 'rho': 3.5, 'AutoRho': {'Enabled': False}}
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert b._dev.uses_fused_rows() and b._fused_ok()
    b.profile(True)
    Y = b.solve()
    cnt = kernel_counts(b)
    assert cnt.get('fused_cols_sm', 0) == 6 and 'sm_solve' not in cnt
    sel = [0, 31]
    ref = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, 64), S[:, :, sel].reshape(512, 512, 1, 2, 1), 0.05,
                         dtype=np.float64, maxiter=6, rel_tol=0.0, rho=3.5, auto_rho=False)
    for j, n in enumerate(sel):
        assert rel_l2(Y[:, :, 0, n], ref['Y'][:, :, 0, j]) < 1e-4, n
    # the objective of the batch is the sum over images: the two checked images' share is
    # consistent with the batch total (every image has the same statistics)
    assert 0.8 < 16.0 * ref['ObjFun'][-1] / b.getitstat().ObjFun[-1] < 1.25


def test_config2_fullsize_fused_properties(gpu_backend):
    """Size-independent properties of the three-launch path at the full config 2 batch (no
    LinSolveCheck: that option switches to the generic chain, see the next test): image
    independence, Parseval consistency of the data-fidelity sum, the l1 sum, determinism."""
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(12345)
    D = rng.randn(8, 8, 64).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(512, 512, 32).astype(np.float32)
    optd = {'MaxMainIter': 4, 'RelStopTol': 0.0, 'rho': 3.5, 'AutoRho': {'Enabled': False}}
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert b._dev.uses_fused_rows() and b._fused_ok()
    b.profile(True)
    Y = b.solve()
    cnt = kernel_counts(b)
    assert cnt.get('fused_cols_sm', 0) == 4 and 'sm_solve' not in cnt
    its = b.getitstat()
    one = cbpdn.ConvBPDN(D, S[:, :, 5], 0.05, cbpdn.ConvBPDN.Options(optd), dimK=0)
    assert one._dev.uses_fused_rows()
    Y1 = one.solve()
    assert rel_l2(Y[:, :, 0, 5], Y1[:, :, 0, 0]) < 1e-6
    rec = b.reconstruct(b.X)
    dfid_spatial = 0.5 * np.sum((rec[:, :, 0, :].astype(np.float64) - S) ** 2)
    assert abs(its.DFid[-1] - dfid_spatial) < 1e-4 * dfid_spatial
    assert abs(its.RegL1[-1] - np.abs(b.X.astype(np.float64)).sum()) < 1e-5 * its.RegL1[-1]
    b2 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert np.array_equal(b2.solve(), Y)


def test_config2_fullsize_linsolvecheck_generic_chain(gpu_backend):
    """The reference's own check on the X-step linear solve (XSlvRelRes < 1e-5,
    tests/admm/test_cbpdn.py:124-139) at the config 2 size.  `LinSolveCheck` needs the
    right-hand side and X in memory, so this runs the generic chain -- and must agree with the
    three-launch path on the iterates."""
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(12345)
    D = rng.randn(8, 8, 64).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(512, 512, 8).astype(np.float32)
    optd = {'MaxMainIter': 4, 'RelStopTol': 0.0, 'rho': 3.5, 'AutoRho': {'Enabled': False}}
    g = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(dict(optd, LinSolveCheck=True)))
    Yg = g.solve()
    assert max(g.getitstat().XSlvRelRes) < 1e-5
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert rel_l2(b.solve(), Yg) < 2e-5


def test_config4_fused_fista_kernels_vs_oracle(gpu_backend):
    """pgm.cbpdn.ConvBPDN 512x512, K=64, N=2, L=500 (the class default), 6 iterations: the
    `pgm_grad_ifft<.,.,64>` / `rows_inv_prox_fwd` / `pgm_fft_momentum<.,.,64,.>` kernels of
    config 4 against the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.pgm import cbpdn as pc
    rng = np.random.RandomState(404)
    D = rng.randn(8, 8, 64).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(512, 512, 2).astype(np.float32)
    for L in (500.0, 20.0):          # the default step and one where X is far from zero
        b = pc.ConvBPDN(D, S, 0.05, pc.ConvBPDN.Options({'MaxMainIter': 6, 'RelStopTol': 0.0,
                                                         'L': L}))
        assert b.dev.uses_fused_rows() and b.dev.uses_fused_pgm() and b._fused_ok()
        X = b.solve()
        ref = orc.pgm_cbpdn(D.reshape(8, 8, 1, 1, 64), S.reshape(512, 512, 1, 2, 1), 0.05,
                            dtype=np.float64, maxiter=6, L=L, rel_tol=0.0)
        assert rel_l2(X, ref['X']) < 1e-4, L
        its = b.getitstat()
        for f in ('ObjFun', 'DFid', 'RegL1', 'Rsdl'):
            assert rel_l2(getattr(its, f), ref[f]) < 1e-4, (L, f)
        assert rel_l2(b.Xf, np.fft.rfftn(ref['X'], axes=(0, 1))) < 1e-4


def test_config5_tiled_dstep_kernels_vs_numpy(gpu_backend):
    """One pgm.ccmod.ConvCnstrMOD iteration at 256x256, K=64, N=4: `ccmod_grad_tiled<.,64>`
    and the register-resident setcoef against the NumPy restatement of pgm/ccmod.py:295-323."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.pgm import ccmod
    H, W, K, N = 256, 256, 64, 4
    rng = np.random.RandomState(505)
    Z = (rng.randn(H, W, 1, N, K) * (rng.rand(H, W, 1, N, K) < 0.02)).astype(np.float32)
    S = rng.randn(H, W, N).astype(np.float32)
    D0 = orc.pcn(rng.randn(H, W, 1, 1, K), (8, 8, K), (H, W)).astype(np.float32)
    L = 14.0 * N * 50
    opt = ccmod.ConvCnstrMOD.Options({'MaxMainIter': 1, 'L': L, 'X0': D0})
    c = ccmod.ConvCnstrMOD(Z, S, (8, 8, K), opt)
    assert c.dev.uses_fused_rows()
    c.solve()
    Zf = np.fft.rfftn(Z.astype(np.float64), axes=(0, 1))
    Sf = np.fft.rfftn(S.astype(np.float64).reshape(H, W, 1, N, 1), axes=(0, 1))
    Df = np.fft.rfftn(D0.astype(np.float64), axes=(0, 1))
    R = np.sum(Zf * Df, axis=4, keepdims=True) - Sf
    G = np.sum(np.conj(Zf) * R, axis=3, keepdims=True)
    # the gradient itself (before the projection normalises it away)
    V = np.fft.irfftn(Df - G / L, (H, W), axes=(0, 1))
    D1 = orc.pcn(V, (8, 8, K), (H, W))
    assert rel_l2(c.getdict(crop=False), D1) < 1e-5
    its = c.getitstat()
    R1 = np.sum(Zf * np.fft.rfftn(D1, axes=(0, 1)), axis=4, keepdims=True) - Sf
    dfid = 0.5 * np.sum(np.fft.irfftn(R1, (H, W), axes=(0, 1)) ** 2)
    assert abs(its.DFid[-1] - dfid) < 1e-5 * dfid
    assert rel_l2(c.Zf, Zf) < 1e-5


def test_config5_dictlearn_k64_vs_reference_fixture(gpu_backend):
    """ConvBPDNDictLearn (xmethod admm, dmethod pgm) 256x256, K=64, N=4, 4 outer iterations
    -- config 5's kernels end to end -- against the float64 run of the reference on the same
    seeded inputs (tests/golden/cbpdndl_config5_n4_f64.npz, oracle/make_golden.py config5)."""
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden('cbpdndl_config5_n4_f64')
    rng = np.random.RandomState(515)
    D0 = rng.randn(8, 8, 64).astype(np.float32)
    S = rng.randn(256, 256, 4).astype(np.float32)
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 4, 'AccurateDFid': True},
                                            xmethod='admm', dmethod='pgm')
    d = cbpdndl.ConvBPDNDictLearn(D0, S, float(g['lmbda']), opt, xmethod='admm', dmethod='pgm')
    assert d.xstep._dev.uses_fused_rows()
    D1 = d.solve()
    assert rel_l2(D1.squeeze(), g['D1'].squeeze()) < 1e-4
    X = d.getcoef()
    assert rel_l2(X[::16, ::16], g['X_sub']) < 1e-4
    assert abs(np.linalg.norm(X.astype(np.float64)) - float(g['X_l2'])) < 1e-4 * float(g['X_l2'])
    its = d.getitstat()
    for f in its._fields:
        if 'it_' + f in g and f not in ('Iter', 'Cnstr'):
            assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f]) < 1e-3, f
    # the constraint violation ||D - Pcn(D)|| is rounding noise in both (1e-15 / 1e-6)
    assert max(its.Cnstr) < 1e-5 and float(np.max(g['it_Cnstr'])) < 1e-12


def test_config3_joint_k128_slabs_vs_oracle(gpu_backend):
    """ConvBPDNJoint, C = 3, K = 128 (two 64-filter slabs in the column pass, joint l1 + l2,1
    epilogue), 256x256, N = 1, 8x8 filters, lambda = 0.1, mu = 0.01 as config 3, 5 iterations
    with default options, against the float64 oracle (25 M elements per array)."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(303)
    H, W, C, N, K = 256, 256, 3, 1, 128
    D = rng.randn(8, 8, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, C, N).astype(np.float32)
    opt = cbpdn.ConvBPDNJoint.Options({'MaxMainIter': 5, 'RelStopTol': 0.0})
    b = cbpdn.ConvBPDNJoint(D, S, 0.1, 0.01, opt)
    assert b._dev.uses_fused_rows() and b._dev.uses_fused_cols()
    Y = b.solve()
    ref = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K), S.reshape(H, W, C, N, 1), 0.1, mu=0.01,
                         dtype=np.float64, maxiter=5, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-4
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-3, f


@pytest.mark.parametrize('K', [32, 128])
def test_config3_joint_at_w512_vs_oracle(gpu_backend, K):
    """Config 3's own instantiations: the W = 512 joint row epilogue
    (rows_inv_post<16, ..., JOINT>) with rows_fwd<16> and the H = 512 column kernels --
    fused_cols at K = 32, the cooperating 64-filter slabs at K = 128 -- on one 512x512 RGB image,
    lambda = 0.1, mu = 0.01, default options, against the float64 oracle (ConvBPDNJoint.ystep
    sporco/admm/cbpdn.py:785-794, prox_sl1l2 sporco/prox/_l21.py:51-88, obfn_reg :798-807).
    K = 128 is 100 M elements per array: the comparison runs blockwise."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(512 + K)
    H, W, C, N = 512, 512, 3, 1
    iters = 5 if K == 32 else 3
    D = rng.randn(8, 8, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, C, N).astype(np.float32)
    opt = cbpdn.ConvBPDNJoint.Options({'MaxMainIter': iters, 'RelStopTol': 0.0})
    b = cbpdn.ConvBPDNJoint(D, S, 0.1, 0.01, opt)
    assert b._dev.uses_fused_rows() and b._dev.uses_fused_cols() and b._fused_ok()
    Y = b.solve()
    ref = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K), S.reshape(H, W, C, N, 1), 0.1, mu=0.01,
                         dtype=np.float64, maxiter=iters, rel_tol=0.0)
    num = den = 0.0
    for h in range(0, H, 64):
        d = Y[h:h + 64].astype(np.float64) - ref['Y'][h:h + 64]
        num += float(np.sum(d * d))
        den += float(np.sum(ref['Y'][h:h + 64] ** 2))
    assert np.sqrt(num / den) < 1e-4
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-3, f


def test_stopping_iteration_fused_f32_vs_reference(gpu_backend):
    """Time-to-tolerance known answer: the sparse-synthesis input of
    bench.make_structured_problem at 512x512, K=64, N=2, lambda 0.01, default options,
    RelStopTol 1e-3.  The fused float32 path must stop within one iteration of the
    reference's own float32 run (tests/golden/admm_tol_config2_n2_f32.npz)
    and land on the same coefficient maps."""
    import bench
    from sporco_amd.admm import cbpdn
    g = load_golden('admm_tol_config2_n2_f32')
    D, S = bench.make_structured_problem(512, 512, 64, 2, 0)
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 1000, 'RelStopTol': 1e-3})
    b = cbpdn.ConvBPDN(D, S, float(g['lmbda']), opt)
    assert b._dev.uses_fused_rows() and b._fused_ok()
    Y = b.solve()
    k_ref = int(g['k_final'])
    assert abs(b.k - k_ref) <= 1, (b.k, k_ref)
    n = min(b.k, k_ref)
    its = b.getitstat()
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(np.asarray(getattr(its, f))[:n], g['it_' + f][:n]) < 2e-3, f
    assert rel_l2(Y[::16, ::16], g['Y_sub']) < 2e-3
