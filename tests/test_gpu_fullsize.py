"""GPU-only tests at BASELINE.json sizes.

The oracle cannot run the full configurations in seconds, so these use
(a) the golden traces the reference produced for config 1 (256x256, K=32, N=1),
(b) the NumPy oracle at a size it finishes quickly (128x128, K=16, N=4), and
(c) size-independent properties at config 2 (512x512, K=64, N=32):
    the reference's own LinSolveCheck assertion (XSlvRelRes < 1e-5,
    tests/admm/test_cbpdn.py:124-139), image independence under fixed rho
    (an image solved inside the batch equals the same image solved alone),
    Parseval consistency of the data-fidelity term, and determinism.
"""

import numpy as np
import pytest

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def normalised_dict(rng, K, dtype=np.float32):
    D = rng.randn(8, 8, K).astype(dtype)
    return D / np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))


@pytest.mark.parametrize('name,dt', [('admm_config1_f32', np.float32),
                                     ('admm_config1_f64', np.float64)])
def test_config1_against_reference_traces(gpu_backend, name, dt):
    from sporco_amd.admm import cbpdn
    g = load_golden(name)
    rng = np.random.RandomState(int(g['seed']))
    D = normalised_dict(rng, 32)
    S = rng.randn(256, 256).astype(np.float32)
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 20, 'RelStopTol': 0.0, 'DataType': dt})
    b = cbpdn.ConvBPDN(D, S, float(g['lmbda']), opt, dimK=0)
    Y = b.solve()
    tol = 1e-9 if dt is np.float64 else 2e-4
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), g['it_' + f]) < tol, f
    assert rel_l2(Y[::16, ::16], g['Y_sub']) < (1e-8 if dt is np.float64 else 5e-4)
    assert abs(np.linalg.norm(Y.astype(np.float64)) - float(g['Y_l2'])) < tol * float(g['Y_l2'])
    if dt is np.float32:
        # 1e-4 bar against the float64 reference run of the same problem
        g64 = load_golden('admm_config1_f64')
        assert rel_l2(Y[::16, ::16], g64['Y_sub']) < 1e-4


def test_oracle_parity_midsize(gpu_backend):
    from sporco_amd.admm import cbpdn
    from oracle import cbpdn_oracle as orc
    rng = np.random.RandomState(7)
    D = normalised_dict(rng, 16)
    S = rng.randn(128, 128, 4).astype(np.float32)
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 30, 'RelStopTol': 0.0})
    b = cbpdn.ConvBPDN(D, S, 0.05, opt)
    Y = b.solve()
    ref = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, 16), S.reshape(128, 128, 1, 4, 1), 0.05,
                         dtype=np.float64, maxiter=30, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-4
    its = b.getitstat()
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-3, f


def test_oracle_parity_joint_and_pgm_midsize(gpu_backend):
    from sporco_amd.admm import cbpdn
    from sporco_amd.pgm import cbpdn as pgm_cbpdn
    from oracle import cbpdn_oracle as orc
    rng = np.random.RandomState(11)
    D = normalised_dict(rng, 16)
    S = rng.randn(96, 80, 3, 2).astype(np.float32)          # C = 3, N = 2, non-square
    opt = cbpdn.ConvBPDNJoint.Options({'MaxMainIter': 25, 'RelStopTol': 0.0})
    b = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, opt)
    Y = b.solve()
    ref = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, 16), S.reshape(96, 80, 3, 2, 1), 0.05, mu=0.02,
                         dtype=np.float64, maxiter=25, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-4
    assert rel_l2(b.getitstat().RegL21, ref['RegL21']) < 1e-3
    opt = pgm_cbpdn.ConvBPDN.Options({'MaxMainIter': 25, 'RelStopTol': 0.0, 'L': 50.0})
    p = pgm_cbpdn.ConvBPDN(D, S, 0.05, opt)
    X = p.solve()
    ref = orc.pgm_cbpdn(D.reshape(8, 8, 1, 1, 16), S.reshape(96, 80, 3, 2, 1), 0.05,
                        dtype=np.float64, maxiter=25, L=50.0, rel_tol=0.0)
    assert rel_l2(X, ref['X']) < 1e-4
    assert rel_l2(p.getitstat().ObjFun, ref['ObjFun']) < 1e-4


def test_config2_fullsize_properties(gpu_backend):
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(12345)
    D = normalised_dict(rng, 64)
    S = rng.randn(512, 512, 32).astype(np.float32)
    optd = {'MaxMainIter': 4, 'RelStopTol': 0.0, 'rho': 3.5, 'AutoRho': {'Enabled': False},
            'LinSolveCheck': True}
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    Y = b.solve()
    its = b.getitstat()
    # the reference's own check on the X-step linear solve
    assert max(its.XSlvRelRes) < 1e-5
    # image independence: image 5 of the batch == the same image solved alone
    one = cbpdn.ConvBPDN(D, S[:, :, 5], 0.05, cbpdn.ConvBPDN.Options(optd), dimK=0)
    Y1 = one.solve()
    assert rel_l2(Y[:, :, 0, 5], Y1[:, :, 0, 0]) < 1e-6
    # Parseval: DFid from the frequency-domain by-product == spatial-domain value
    rec = b.reconstruct(b.X)
    dfid_spatial = 0.5 * np.sum((rec[:, :, 0, :].astype(np.float64) - S) ** 2)
    assert abs(its.DFid[-1] - dfid_spatial) < 1e-4 * dfid_spatial
    # l1 term and sparsity sanity
    assert abs(its.RegL1[-1] - np.abs(b.X.astype(np.float64)).sum()) < 1e-5 * its.RegL1[-1]
    # determinism: a second solver object reproduces the iterates bit for bit
    b2 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert np.array_equal(b2.solve(), Y)


def test_config3_shard_fullsize(gpu_backend):
    """Config 3 as one of its 8 GPUs sees it: ConvBPDNJoint, 512x512 RGB, K = 128, 32 of
    the 256 images -- 3.2e9 elements (12.9 GB) per X-sized array, i.e. past 2^31 elements.
    The slab column kernels + fused row passes against the generic kernel chain."""
    import os
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(3)
    H, C, N, K = 512, 3, 32, 128
    D = rng.randn(8, 8, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, H, C, N).astype(np.float32)

    def run(unfused):
        if unfused:
            os.environ['SPORCO_AMD_UNFUSED'] = '1'
        try:
            opt = cbpdn.ConvBPDNJoint.Options({'MaxMainIter': 4, 'RelStopTol': 0.0})
            b = cbpdn.ConvBPDNJoint(D, S, 0.1, 0.02, opt)
        finally:
            os.environ.pop('SPORCO_AMD_UNFUSED', None)
        Y = b.solve()
        return Y, b.getitstat(), bool(b._dev.uses_fused_cols())

    Y, its, fused = run(False)
    Y0, its0, fused0 = run(True)
    assert fused and not fused0
    num = den = 0.0
    for h in range(0, H, 32):         # blockwise: no 26 GB float64 temporaries
        d = Y[h:h + 32].astype(np.float64) - Y0[h:h + 32]
        num += float(np.sum(d * d))
        den += float(np.sum(Y0[h:h + 32].astype(np.float64) ** 2))
    assert np.sqrt(num / den) < 2e-5
    for f in ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(np.asarray(getattr(its, f), float), np.asarray(getattr(its0, f), float)) < 1e-5, f
    # every image's last filter was reached (the far end of the 12.9 GB arrays is not zero)
    assert np.count_nonzero(Y[-1, -8:, :, -1, -1]) + np.count_nonzero(Y[0, :8, :, -1, -1]) > 0


def _blockwise_rel_l2(a, b, step=64):
    num = den = 0.0
    for h in range(0, a.shape[0], step):
        d = a[h:h + step].astype(np.float64) - b[h:h + step]
        num += float(np.sum(d * d))
        den += float(np.sum(np.asarray(b[h:h + step], dtype=np.float64) ** 2))
    return np.sqrt(num / den)


def test_config2_every_image_vs_oracle(gpu_backend):
    """ALL 32 images of the benchmarked problem (512x512, K=64, N=32: bench.make_problem), 6
    iterations at fixed rho, each against the float64 oracle run on that image alone (at fixed
    rho the iteration is independent per image, sporco/admm/admm.py:331-367): an external witness
    for every image slot of the batch, not only the first and the last.  About two minutes of
    host time."""
    import bench
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    D, S = bench.make_problem(512, 512, 64, 32, 0)
    optd = {'MaxMainIter': 6, 'RelStopTol': 0.0, 'rho': 3.5, 'AutoRho': {'Enabled': False}}
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert b._dev.uses_fused_rows() and b._fused_ok()
    Y = b.solve()
    # (the 32 oracle runs are independent: side by side on the host's cores, tests/_oracle_pool.py)
    from _oracle_pool import oracle_solves
    kw = dict(dtype=np.float64, maxiter=6, rel_tol=0.0, rho=3.5, auto_rho=False)
    jobs = [((D.reshape(8, 8, 1, 1, 64), S[:, :, n].reshape(512, 512, 1, 1, 1), 0.05), kw,
             {'Y': ('Y', (slice(None), slice(None), 0, 0), Y[:, :, 0, n])}) for n in range(32)]
    worst = 0.0
    obj = 0.0
    for n, (errs, objn) in enumerate(oracle_solves(jobs)):
        assert errs['Y'] < 1e-4, (n, errs)
        worst = max(worst, errs['Y'])
        obj += objn
    # the batch objective is the sum of the images' (lambda ||x||_1 + 1/2 ||Dx - s||^2 are sums)
    assert abs(obj - b.getitstat().ObjFun[-1]) < 1e-4 * obj
    print('config 2, 32 images vs oracle: worst rel l2 %.2e' % worst)


def test_config3_shard_spread_images_vs_oracle(gpu_backend):
    """The config-3 shard (ConvBPDNJoint, 512x512 RGB, K = 128, N = 32: 3.2e9 elements per array,
    index arithmetic past 2^31) at fixed rho, 3 iterations: images 0, 10, 21 and 31 -- the first and
    the last, i.e. both ends of the 12.9 GB arrays, and two in between -- against the float64
    oracle run on each image alone, over all 128 filters (both 64-filter slabs of the column pass).
    The l2,1 term couples the channels of one image only (sporco/admm/cbpdn.py:785-794)."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(3)
    H, C, N, K = 512, 3, 32, 128
    D = rng.randn(8, 8, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, H, C, N).astype(np.float32)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0, 'rho': 6.0, 'AutoRho': {'Enabled': False}}
    b = cbpdn.ConvBPDNJoint(D, S, 0.1, 0.02, cbpdn.ConvBPDNJoint.Options(optd))
    assert b._dev.uses_fused_rows() and b._dev.uses_fused_cols() and b._fused_ok()
    Y = b.solve()
    from _oracle_pool import oracle_solves
    kw = dict(mu=0.02, dtype=np.float64, maxiter=3, rel_tol=0.0, rho=6.0, auto_rho=False)
    images = (0, 10, 21, 31)
    jobs = []
    for n in images:
        checks = {'Y': ('Y', (slice(None), slice(None), slice(None), 0), Y[:, :, :, n])}
        # both slabs and both ends of the filter axis individually
        for k in (0, 63, 64, 127):
            checks['Y%d' % k] = ('Y', (slice(None), slice(None), slice(None), 0, k), Y[:, :, :, n, k])
        jobs.append(((D.reshape(8, 8, 1, 1, K), S[:, :, :, n].reshape(H, H, C, 1, 1), 0.1), kw, checks))
    for n, (errs, _) in zip(images, oracle_solves(jobs)):
        assert errs['Y'] < 1e-4, (n, errs)
        assert max(errs['Y%d' % k] for k in (0, 63, 64, 127)) < 2e-4, (n, errs)
