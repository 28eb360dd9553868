"""Online dictionary learning (sporco_amd.dictlrn.onlinecdl.OnlineConvBPDNDictLearn) against
fixtures produced by the unmodified reference (oracle/make_golden.py gen_online): dictionary
after every solve() call and the IterationStats rows.  float64 1e-9; float32 against the
reference's own float32 run 2e-3 (30 cold-started ADMM iterations per call, then the SGD step)."""

import numpy as np
import pytest

from conftest import load_golden, rel_l2

FIELDS = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho', 'Cnstr', 'DeltaD', 'Eta')


@pytest.mark.parametrize('name,dt,tol', [('onlinecdl_f64', np.float64, 1e-9),
                                         ('onlinecdl_f32', np.float32, 2e-3),
                                         ('onlinecdl_batch_f64', np.float64, 1e-9)])
def test_golden_traces(backend, name, dt, tol):
    from sporco_amd.dictlrn import onlinecdl
    g = load_golden(name)
    S = g['S']
    if 'batch' in name:
        optd, dimK = {'CBPDN': {'MaxMainIter': 20}}, 1
    else:
        optd, dimK = {'eta_a': 8.0, 'eta_b': 4.0, 'ZeroMean': dt is np.float64, 'DataType': dt,
                      'CBPDN': {'MaxMainIter': 30}}, 0
    b = onlinecdl.OnlineConvBPDNDictLearn(g['D0'], float(g['lmbda']),
                                          onlinecdl.OnlineConvBPDNDictLearn.Options(optd),
                                          dimK=dimK)
    for i in range(S.shape[-1]):
        D = b.solve(S[..., i].astype(dt))
        assert D.shape == g['Ds'][i].shape and D.dtype == dt
        assert rel_l2(D, g['Ds'][i]) < tol, i
    assert b.j == S.shape[-1]
    its = b.getitstat()
    assert its._fields == ('Iter',) + FIELDS + ('Time',)
    for f in FIELDS:
        assert rel_l2(np.asarray(getattr(its, f), dtype=float), g['it_' + f]) < tol, f
    assert b.getcoef().shape[:2] == S.shape[:2]


def test_masked_golden_traces(backend):
    """OnlineConvBPDNMaskDictLearn (onlinecdl.py:464-600): binary masks and a non-binary
    weighting (the gradient weights the residual by W once, not W^2)."""
    from sporco_amd.dictlrn import onlinecdl
    g = load_golden('onlinecdl_mask_f64')
    cls = onlinecdl.OnlineConvBPDNMaskDictLearn
    assert cls.Options()['CBPDN', 'MaxMainIter'] == 1000 and cls.Options()['CBPDN', 'rho'] == 1.0
    b = cls(g['D0'], float(g['lmbda']),
            cls.Options({'eta_a': 8.0, 'eta_b': 4.0, 'CBPDN': {'MaxMainIter': 30}}), dimK=0)
    for i in range(g['S'].shape[-1]):
        D = b.solve(g['S'][..., i], g['W'][..., i])
        assert rel_l2(D, g['Ds'][i]) < 1e-9, i
    its = b.getitstat()
    for f in FIELDS:
        assert rel_l2(np.asarray(getattr(its, f), dtype=float), g['it_' + f]) < 1e-9, f


def test_masked_step_at_image_size_float32(backend):
    """One masked online step at 256 x 256 in float32 (X-step on the generic chain, coefficient
    spectra through the tile-major transform and back for the masked gradient) against the
    float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.dictlrn import onlinecdl
    rng = np.random.RandomState(11)
    H, K = 256, 4
    D0 = rng.randn(8, 8, K)
    S = rng.randn(H, H).astype(np.float32)
    W = (rng.rand(H, H) > 0.3).astype(np.float32)
    cls = onlinecdl.OnlineConvBPDNMaskDictLearn
    o = cls(D0, 0.1, cls.Options({'DataType': np.float32, 'CBPDN': {'MaxMainIter': 3}}), dimK=0)
    D1 = o.solve(S, W)
    sh = (H, H, 1, 1, 1)
    ref = orc.online_cdl(D0, [S.reshape(sh).astype(np.float64)], 0.1, xstep_iter=3,
                         masks=[W.reshape(sh).astype(np.float64)])
    assert rel_l2(D1, ref['Ds'][0]) < 1e-5
    assert rel_l2(o.getitstat().Cnstr, ref['Cnstr']) < 1e-5


def test_surface(backend):
    from sporco_amd.dictlrn import onlinecdl
    cls = onlinecdl.OnlineConvBPDNDictLearn
    opt = cls.Options()
    assert opt['CBPDN', 'MaxMainIter'] == 100 and opt['CBPDN', 'AutoRho', 'Period'] == 10
    assert opt['eta_a'] == 10.0 and opt['eta_b'] == 5.0
    D0 = np.random.RandomState(0).randn(4, 4, 3)
    with pytest.raises(TypeError):
        cls(D0, 0.1, {'eta_a': 1.0})
    with pytest.raises(ValueError):
        cls(D0, 0.1, cls.Options({'CUDA_CBPDN': True}))
    b = cls(D0, 0.1, cls.Options({'Verbose': False, 'CBPDN': {'MaxMainIter': 2}}), dimK=0)
    assert abs(np.linalg.norm(b.getdict()[..., 0]) - 1.0) < 1e-12     # Pcn of D0
    # images of different sizes may follow each other (init_vars, onlinecdl.py:244-263)
    for n in (16, 20):
        D = b.solve(np.random.RandomState(n).randn(n, n))
    assert D.shape == (4, 4, 1, 1, 3) and b.j == 2
    assert b.getitstat().Eta[-1] == 10.0 / 6.0


@pytest.mark.gpu
def test_image_sized_step_against_oracle(gpu_backend):
    """One online step at 256 x 256 with 32 filters, float32, on the fused X-step kernels and
    the tile-major gradient, against the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.dictlrn import onlinecdl
    rng = np.random.RandomState(5)
    H, K = 256, 32
    D0 = rng.randn(8, 8, K)
    imgs = [rng.randn(H, H).astype(np.float32) for _ in range(2)]
    cls = onlinecdl.OnlineConvBPDNDictLearn
    b = cls(D0, 0.2, cls.Options({'DataType': np.float32, 'CBPDN': {'MaxMainIter': 10}}),
            dimK=0)
    Ds = [b.solve(s).copy() for s in imgs]
    assert b._xstep._dev.uses_fused_rows()
    ref = orc.online_cdl(D0, [s.reshape(H, H, 1, 1, 1) for s in imgs], 0.2, dtype=np.float64,
                         xstep_iter=10)
    assert rel_l2(np.stack(Ds), ref['Ds']) < 1e-4
    its = b.getitstat()
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Cnstr', 'DeltaD'):
        assert rel_l2(np.asarray(getattr(its, f), dtype=float), ref[f]) < 1e-4, f
