"""Pin the NumPy oracle (oracle/cbpdn_oracle.py) to the reference itself.

The fixtures in tests/golden/ were produced by oracle/make_golden.py from the
unmodified reference (bwohlberg/sporco).  Every oracle function used by the
GPU parity tests is checked here on CPU.  The path is floating point, so the
pins are tolerances (float64 cases: 1e-10 or tighter; float32: 1e-4).
"""

import numpy as np
import pytest

from conftest import load_golden, rel_l2
from oracle import cbpdn_oracle as orc


def to5d(D, S, dimK=None):
    """Internal 5-D layout (cnvrep.py:186-198); a 4-D D is a multi-channel dictionary."""
    if D.ndim == 4:
        D5 = D.reshape(D.shape[0], D.shape[1], D.shape[2], 1, D.shape[3])
        N = S.shape[3] if S.ndim == 4 else 1
        return D5, S.reshape(S.shape[0], S.shape[1], S.shape[2], N, 1)
    D5 = D.reshape(D.shape[0], D.shape[1], 1, 1, D.shape[-1])
    rdim = S.ndim - 2
    if dimK is None:
        dimC, dimKk = (0, 0) if rdim == 0 else ((0, 1) if rdim == 1 else (1, 1))
    else:
        dimKk, dimC = dimK, rdim - dimK
    C = S.shape[2] if dimC else 1
    N = S.shape[2 + dimC] if dimKk else 1
    return D5, S.reshape(S.shape[0], S.shape[1], C, N, 1)


def test_primitives():
    g = load_golden('primitives')
    x = orc.solvedbi_sm(g['sm_ah'], float(g['sm_rho']), g['sm_b'])
    assert rel_l2(x, g['sm_x']) < 1e-13
    assert rel_l2(orc.solvedbi_sm_c(g['sm_ah'], np.conj(g['sm_ah']),
                                    float(g['sm_rho'])), g['sm_c']) < 1e-13
    assert rel_l2(orc.inner(g['sm_ah'], g['sm_b'], axis=4), g['inner_ab']) < 1e-13
    assert rel_l2(orc.rfftn2(g['fft_a']), g['fft_af']) < 1e-13
    assert rel_l2(orc.irfftn2(g['fft_af'], (12, 9)), g['fft_ar']) < 1e-13
    assert rel_l2(orc.rfftn2(g['fft_a2']), g['fft_a2f']) < 1e-6
    assert orc.rfftn2(g['fft_a2']).dtype == np.complex64
    assert rel_l2(orc.rfftn2(g['fft_d'], (12, 9)), g['fft_df']) < 1e-13
    assert abs(orc.rfl2norm2(g['fft_af'], g['fft_a'].shape) - g['nrm_odd']) \
        < 1e-12 * g['nrm_odd']
    assert abs(orc.rfl2norm2(g['fft_a2f'], g['fft_a2'].shape) - g['nrm_even']) \
        < 1e-6 * g['nrm_even']
    a = float(g['prox_alpha'])
    assert rel_l2(orc.prox_l1(g['prox_v'], a), g['prox_l1']) == 0.0
    assert rel_l2(orc.prox_l1(g['prox_v'], a * g['prox_w']), g['prox_l1w']) == 0.0
    assert rel_l2(orc.prox_l2(g['prox_v'], 0.9, axis=2), g['prox_l2']) < 1e-15
    assert rel_l2(orc.prox_sl1l2(g['prox_v'], a, 0.9, axis=2),
                  g['prox_sl1l2']) < 1e-15
    assert rel_l2(orc.prox_sl1l2(g['prox_vz'], a, 0.9, axis=2),
                  g['prox_sl1l2_z']) < 1e-15


ADMM_CASES = {
    'admm_default_f64': dict(maxiter=30),
    'admm_default_f32': dict(maxiter=30, dtype=np.float32),
    'admm_fixedrho_f64': dict(maxiter=25, rho=2.0, rlx=1.0, auto_rho=False),
    'admm_autorho_std_f64': dict(maxiter=30, rho_period=3, auto_scaling=False,
                                 rho_tau=2.0, rho_mu=1.5, std_residuals=True,
                                 abs_tol=1e-6),
    'admm_odd_nonneg_nobndry_f64': dict(maxiter=30, nonneg=True, nobndry=True),
    'admm_l1weight_auxvar_f64': dict(maxiter=25, gevaly=True, fevalx=False,
                                     _wl1='optarr_L1Weight'),
    'admm_l1weight_spatial_f64': dict(maxiter=20, _wl1='optarr_L1Weight'),
    'admm_multichan_f64': dict(maxiter=25),
    'admm_joint_f64': dict(maxiter=25),
    'admm_joint_f32': dict(maxiter=25, dtype=np.float32),
    'admm_joint_l21weight_f64': dict(maxiter=20, nonneg=True,
                                     _wl21='optarr_L21Weight'),
    'admm_warmstart_f64': dict(maxiter=15, _y0='optarr_Y0', _u0='optarr_U0'),
}


@pytest.mark.parametrize('name', sorted(ADMM_CASES))
def test_admm_traces(name):
    g = load_golden(name)
    kw = dict(ADMM_CASES[name])
    dtype = kw.pop('dtype', np.float64)
    tol = 1e-9 if dtype == np.float64 else 2e-4
    if '_wl1' in kw:
        kw['wl1'] = g[kw.pop('_wl1')]
    if '_wl21' in kw:
        kw['wl21'] = g[kw.pop('_wl21')]
    if '_y0' in kw:
        kw['Y0'] = g[kw.pop('_y0')]
        kw['U0'] = g[kw.pop('_u0')]
    dimK = None if int(g['dimK']) < 0 else int(g['dimK'])
    D5, S5 = to5d(g['D'], g['S'], dimK)
    mu = None if float(g['mu']) < 0 else float(g['mu'])
    r = orc.admm_cbpdn(D5, S5, float(g['lmbda']), mu=mu, dtype=dtype, **kw)
    assert r['iters'] == int(g['k_final'])
    assert rel_l2(r['Y'], g['Y']) < tol
    assert rel_l2(r['U'], g['U']) < tol
    assert rel_l2(r['X'], g['X']) < tol
    for key in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl',
                'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key
    if mu is not None:
        assert rel_l2(r['RegL21'], g['it_RegL21']) < tol
    assert rel_l2(orc.reconstruct(r['Df'], r['Y'], S5.shape[:2]),
                  g['recon']) < tol


GRADREG_CASES = {
    'admm_gradreg_f64': dict(maxiter=30),
    'admm_gradreg_f32': dict(maxiter=30, dtype=np.float32),
    'admm_gradreg_weights_f64': dict(maxiter=25, rho=1.5, auto_rho=False,
                                     nonneg=True, _wg='optarr_GradWeight'),
    'admm_gradreg_auxvar_f64': dict(maxiter=20, gevaly=True, fevalx=False,
                                    _wg='optarr_GradWeight'),
    # multi-channel dictionary: the iterated solve with the diagonal (cbpdn.py:1181-1184)
    'admm_gradreg_mcdict_f64': dict(maxiter=20, _wg='optarr_GradWeight'),
    'admm_gradreg_mcdict_f32': dict(maxiter=20, dtype=np.float32),
}


@pytest.mark.parametrize('name', sorted(GRADREG_CASES))
def test_gradreg_traces(name):
    """ConvBPDNGradReg restatement (solvedbd_sm, gradient filters, RegGrad)."""
    g = load_golden(name)
    kw = dict(GRADREG_CASES[name])
    dtype = kw.pop('dtype', np.float64)
    tol = 1e-9 if dtype == np.float64 else 2e-4
    if '_wg' in kw:
        kw['grad_weight'] = g[kw.pop('_wg')]
    D5, S5 = to5d(g['D'], g['S'])
    r = orc.admm_cbpdn(D5, S5, float(g['lmbda']), grad_mu=float(g['mu']),
                       dtype=dtype, **kw)
    assert rel_l2(orc.gradient_filters_ghg(S5.shape[:2], np.float64)
                  * (kw.get('grad_weight', 1.0)), g['GHGf']) < 1e-6
    assert r['iters'] == int(g['k_final'])
    for key in ('Y', 'U', 'X', 'Xf'):
        assert rel_l2(r[key], g[key]) < tol, key
    for key in ('ObjFun', 'DFid', 'RegL1', 'RegGrad', 'PrimalRsdl', 'DualRsdl',
                'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key


AMS_CASES = {
    'ams_cbpdn_f64': dict(maxiter=25),
    'ams_cbpdn_f32': dict(maxiter=25, dtype=np.float32),
    'ams_cbpdn_bcast_nonneg_f64': dict(maxiter=20, nonneg=True, nobndry=True,
                                       gevaly=True, fevalx=False),
    'ams_gradreg_f64': dict(maxiter=20, _wg='optarr_GradWeight', _gradreg=True),
    'ams_joint_f64': dict(maxiter=20, _joint=True),
    'ams_cbpdn_mcdict_f64': dict(maxiter=20),
    'ams_cbpdn_mcdict_bcast_f64': dict(maxiter=20, nonneg=True, gevaly=True, fevalx=False),
    'ams_gradreg_mcdict_f64': dict(maxiter=15, _gradreg=True),
}


def ams_inputs(g):
    """Dictionary with the impulse filter appended (cbpdn.py:2345-2353) and the
    internal 5-D arrays."""
    D = g['D']
    if D.ndim == 4:      # multi-channel dictionary: one impulse per channel (:2339-2346)
        Cd = D.shape[2]
        imp = np.zeros(D.shape[:2] + (Cd, Cd))
        for c in range(Cd):
            imp[0, 0, c, c] = 1.0
    else:
        imp = np.zeros(D.shape[:2] + (1,))
        imp[0, 0] = 1.0
    D5, S5 = to5d(np.concatenate((D, imp), axis=D.ndim - 1), g['S'])
    return D5, S5


@pytest.mark.parametrize('name', sorted(AMS_CASES))
def test_ams_traces(name):
    """AddMaskSim restatement (masked impulse slice in the y step, regularisers
    blind to it)."""
    g = load_golden(name)
    kw = dict(AMS_CASES[name])
    dtype = kw.pop('dtype', np.float64)
    tol = 1e-9 if dtype == np.float64 else 2e-4
    D5, S5 = ams_inputs(g)
    if kw.pop('_gradreg', False):
        kw['grad_mu'] = float(g['mu'])
        if '_wg' in kw:
            kw['grad_weight'] = g[kw.pop('_wg')]
    if kw.pop('_joint', False):
        kw['mu'] = float(g['mu'])
    r = orc.admm_cbpdn(D5, S5, float(g['lmbda']), dtype=dtype, ams_mask=g['Wint'], **kw)
    assert r['iters'] == int(g['k_final'])
    for key in ('Y', 'U', 'X'):
        assert rel_l2(r[key], g[key]) < tol, key
    fields = ['ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal',
              'EpsDual', 'Rho']
    fields += ['RegGrad'] if 'grad_mu' in kw else []
    fields += ['RegL21'] if 'mu' in kw else []
    for key in fields:
        assert rel_l2(r[key], g['it_' + key]) < tol, key
    # AddMaskSim.reconstruct / getcoef drop the impulse slice (cbpdn.py:2431-2477)
    ni = D5.shape[2] if D5.shape[2] > 1 else 1      # impulse filters appended
    assert rel_l2(orc.reconstruct(r['Df'][..., :-ni], r['Y'][..., :-ni], S5.shape[:2]),
                  g['recon']) < tol


MCDICT_CASES = {
    'admm_mcdict_f64': dict(maxiter=25),
    'admm_mcdict_f32': dict(maxiter=25, dtype=np.float32),
    'admm_mcdict_single_nonneg_f64': dict(maxiter=20, nonneg=True, gevaly=True, fevalx=False,
                                          rho=2.0, auto_rho=False),
}


def mcdict_inputs(g):
    """Internal layout for a multi-channel dictionary (cnvrep.py:186-194): D
    (dH, dW, C, 1, K), S (H, W, C, N, 1), coefficient maps (H, W, 1, N, K)."""
    D, S = g['D'], g['S']
    D5 = D.reshape(D.shape[0], D.shape[1], D.shape[2], 1, D.shape[3])
    N = S.shape[3] if S.ndim == 4 else 1
    return D5, S.reshape(S.shape[0], S.shape[1], S.shape[2], N, 1)


def test_solvemdbi_ism():
    g = load_golden('solvemdbi_ism')
    x = orc.solvemdbi_ism(g['ah'], float(g['rho']), g['b'], 4, 2)
    assert rel_l2(x, g['x']) < 1e-13
    # it solves the system it claims to solve
    a = np.conj(g['ah'])
    ax = float(g['rho']) * x + np.sum(a * orc.inner(g['ah'], x, axis=4), axis=2, keepdims=True)
    assert rel_l2(ax, g['b']) < 1e-12


@pytest.mark.parametrize('name', sorted(MCDICT_CASES))
def test_mcdict_traces(name):
    g = load_golden(name)
    kw = dict(MCDICT_CASES[name])
    dtype = kw.pop('dtype', np.float64)
    # (float32: two float32 evaluations of 25 adaptive-rho iterations of the C-term
    # iterated solve; observed 2.0e-4)
    tol = 1e-9 if dtype == np.float64 else 5e-4
    D5, S5 = mcdict_inputs(g)
    r = orc.admm_cbpdn(D5, S5, float(g['lmbda']), dtype=dtype, **kw)
    assert r['iters'] == int(g['k_final'])
    for key in ('Y', 'U', 'X', 'Xf'):
        assert r[key].shape == g[key].shape
        assert rel_l2(r[key], g[key]) < tol, key
    for key in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal',
                'EpsDual', 'Rho'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key
    assert rel_l2(orc.reconstruct(r['Df'], r['Y'], S5.shape[:2]).squeeze(),
                  g['recon'].squeeze()) < tol


CNS_CASES = {
    'ccmod_cns_f64': dict(maxiter=20),
    'ccmod_cns_f32': dict(maxiter=20, dtype=np.float32),
    'ccmod_cns_autorho_zm_f64': dict(maxiter=25, zero_mean=True, rho=2.0, rlx=1.5,
                                     auto_rho=True, rho_period=2, rho_tau=2.0, rho_mu=1.2,
                                     auto_scaling=True, rho_xi=1.0),
    'ccmod_cns_y0_f64': dict(maxiter=10, _y0=True),
}


@pytest.mark.parametrize('name', sorted(CNS_CASES))
def test_ccmod_consensus_traces(name):
    """ConvCnstrMOD_Consensus restatement (per-image Sherman-Morrison, mean + Pcn, consensus
    residuals)."""
    g = load_golden(name)
    kw = dict(CNS_CASES[name])
    dtype = kw.pop('dtype', np.float64)
    tol = 1e-9 if dtype == np.float64 else 2e-4
    if kw.pop('_y0', False):
        kw['Y0'] = g['Y0']
    S = g['S']
    r = orc.admm_ccmod_cns(g['Z'], S.reshape(S.shape[0], S.shape[1], 1, S.shape[2], 1),
                           tuple(int(v) for v in g['dsz']), dtype=dtype, **kw)
    if 'k_final' in g:
        assert r['iters'] == int(g['k_final'])
    assert rel_l2(r['Y'], g['Y']) < tol
    assert rel_l2(r['U'], g['U']) < tol
    assert rel_l2(r['D'], g['D']) < tol
    for key in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key
    assert np.max(np.abs(r['Cnstr'] - g['it_Cnstr'])) < 1e-6


EQ_CASES = {
    'f64': dict(maxiter=20),
    'f32': dict(maxiter=20, dtype=np.float32),
    'fixedrho_zm_chk_f64': dict(maxiter=20, rho=5.0, auto_rho=False, zero_mean=True,
                                lin_solve_check=True, rlx=1.5),
    'auxobj_y0_f64': dict(maxiter=12, aux_var_obj=True, _y0=True, rho_period=3, rho_tau=2.0,
                          auto_scaling=False, rho_mu=1.5),
}


@pytest.mark.parametrize('method', ['ism', 'cg'])
@pytest.mark.parametrize('case', sorted(EQ_CASES))
def test_ccmod_ism_cg_traces(method, case):
    """ConvCnstrMOD_IterSM / ConvCnstrMOD_CG restatement (one dictionary copy, X-step by
    iterated Sherman-Morrison over the images or warm-started conjugate gradients)."""
    g = load_golden('ccmod_%s_%s' % (method, case))
    kw = dict(EQ_CASES[case])
    dtype = kw.pop('dtype', np.float64)
    tol = 1e-8 if dtype == np.float64 else 5e-4
    if kw.pop('_y0', False):
        kw['Y0'] = g['Y0']
    if method == 'cg' and 'fixedrho' in case:
        kw.update(cg_tol=1e-9, cg_maxiter=500)
        tol = 1e-7
    elif method == 'cg':
        # CG run to the default 1e-3 stopping tolerance amplifies the rounding of the operator
        # (einsum against broadcast-multiply-sum) to ~1e-5 in the iterate: same stopping rule
        # and iteration flags, looser comparison of the values
        tol = 1e-4 if dtype == np.float64 else 2e-3
    S = g['S']
    r = orc.admm_ccmod_eq(g['Z'], S.reshape(S.shape[0], S.shape[1], 1, S.shape[2], 1),
                          tuple(int(v) for v in g['dsz']), method=method, dtype=dtype, **kw)
    assert r['iters'] == int(g['k_final'])
    for key in ('Y', 'U', 'X', 'D'):
        assert rel_l2(r[key], g[key]) < tol, key
    for key in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key
    assert np.max(np.abs(r['Cnstr'] - g['it_Cnstr'])) < max(10 * tol, 1e-6)
    if 'chk' in case:
        assert np.max(np.abs(r['XSlvRelRes'] - g['it_XSlvRelRes'])) < 1e-6
    if method == 'cg':
        assert np.array_equal(r['XSlvCGIt'], g['it_XSlvCGIt'])


@pytest.mark.parametrize('name,dtype,tol', [('onlinecdl_f64', np.float64, 1e-9),
                                            ('onlinecdl_f32', np.float32, 1e-3),
                                            ('onlinecdl_batch_f64', np.float64, 1e-9)])
def test_online_cdl_traces(name, dtype, tol):
    """OnlineConvBPDNDictLearn restatement: cold-started X-step per batch, one projected SGD
    step on the dictionary."""
    g = load_golden(name)
    S = g['S']
    if 'batch' in name:
        batches = [S[..., i].reshape(S.shape[0], S.shape[1], 1, S.shape[2], 1)
                   for i in range(S.shape[-1])]
        kw = dict(xstep_iter=20)
    else:
        batches = [S[..., i].reshape(S.shape[0], S.shape[1], 1, 1, 1)
                   for i in range(S.shape[-1])]
        kw = dict(xstep_iter=30, eta_a=8.0, eta_b=4.0, zero_mean=dtype == np.float64)
    r = orc.online_cdl(g['D0'], batches, float(g['lmbda']), dtype=dtype, **kw)
    assert rel_l2(r['Ds'], g['Ds']) < tol
    for key in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho', 'Cnstr', 'DeltaD',
                'Eta'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key


MDCPL_CASES = {
    'maskdcpl_f64': dict(maxiter=30),
    'maskdcpl_f32': dict(maxiter=30, dtype=np.float32),
    'maskdcpl_autorho_opts_f64': dict(maxiter=30, rho=2.0, rlx=1.5, nonneg=True, nobndry=True,
                                      aux_var_obj=True, lin_solve_check=True, auto_rho=True,
                                      rho_period=3, rho_tau=2.0, rho_mu=1.2, auto_scaling=True,
                                      _wl1=True),
    'maskdcpl_multichan_f64': dict(maxiter=20),
}


def maskdcpl_inputs(g):
    """5-D D, S and mask (cnvrep.mskWshape: a mask of the signal's shape, or one image-shaped
    mask broadcast over channels and images)."""
    S = g['S']
    D5 = g['D'].reshape(g['D'].shape[:2] + (1, 1, g['D'].shape[2]))
    S5 = S.reshape(S.shape[:2] + ((1, S.shape[2], 1) if S.ndim == 3 else S.shape[2:] + (1,)))
    Wm = g['W']
    W5 = Wm.reshape(S5.shape) if Wm.ndim == S.ndim else Wm.reshape(Wm.shape + (1, 1, 1))
    return D5, S5, W5


@pytest.mark.parametrize('name', sorted(MDCPL_CASES))
def test_maskdcpl_traces(name):
    """ConvBPDNMaskDcpl restatement: two-block constraint, rho-free X-step, dual residual from
    the dual variable."""
    g = load_golden(name)
    kw = dict(MDCPL_CASES[name])
    dtype = kw.pop('dtype', np.float64)
    tol = 1e-9 if dtype == np.float64 else 5e-4
    if kw.pop('_wl1', False):
        kw['wl1'] = g['wl1']
    D5, S5, W5 = maskdcpl_inputs(g)
    r = orc.admm_cbpdn_maskdcpl(D5, S5, float(g['lmbda']), W5, dtype=dtype, **kw)
    assert r['iters'] == int(g['k_final'])
    K = D5.shape[-1]
    assert rel_l2(r['Y1'], g['Y1']) < tol and rel_l2(r['X'], g['X']) < tol
    assert rel_l2(r['Y0'], g['Y'][..., :1]) < tol and rel_l2(r['Y1'], g['Y'][..., 1:]) < tol
    assert rel_l2(r['U0'], g['U'][..., :1]) < tol and rel_l2(r['U1'], g['U'][..., 1:]) < tol
    assert g['Y'].shape[-1] == K + 1
    for key in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual',
                'Rho'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key
    if 'lin_solve_check' in kw:
        assert np.max(np.abs(r['XSlvRelRes'] - g['it_XSlvRelRes'])) < 1e-12


def test_online_masked_cdl_traces():
    """OnlineConvBPDNMaskDictLearn restatement: mask-decoupling X-step, gradient residual
    weighted by the mask once."""
    g = load_golden('onlinecdl_mask_f64')
    S, Wm = g['S'], g['W']
    sh = (S.shape[0], S.shape[1], 1, 1, 1)
    n = S.shape[-1]
    r = orc.online_cdl(g['D0'], [S[..., i].reshape(sh) for i in range(n)], float(g['lmbda']),
                       dtype=np.float64, eta_a=8.0, eta_b=4.0, xstep_iter=30,
                       masks=[Wm[..., i].reshape(sh) for i in range(n)])
    assert rel_l2(r['Ds'], g['Ds']) < 1e-9
    for key in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho', 'Cnstr', 'DeltaD',
                'Eta'):
        assert rel_l2(r[key], g['it_' + key]) < 1e-9, key


CCMODMD_CASES = {
    'f64': dict(maxiter=20),
    'f32': dict(maxiter=20, dtype=np.float32),
    'opts_f64': dict(maxiter=20, rho=3.0, rlx=1.5, zero_mean=True, lin_solve_check=True,
                     aux_var_obj=True, auto_rho=True, rho_period=3, rho_tau=2.0, rho_mu=1.2,
                     auto_scaling=True),
}


@pytest.mark.parametrize('method', ['ism', 'cg'])
@pytest.mark.parametrize('case', sorted(CCMODMD_CASES))
def test_ccmod_maskdcpl_traces(method, case):
    """ConvCnstrMODMaskDcpl_IterSM / _CG restatement (two-block constraint, rho-free X-step
    over the images, dual residual from the dual variable)."""
    g = load_golden('ccmodmd_%s_%s' % (method, case))
    kw = dict(CCMODMD_CASES[case])
    dtype = kw.pop('dtype', np.float64)
    tol = 1e-8 if dtype == np.float64 else 1e-3
    if method == 'cg':
        kw.update(cg_tol=1e-9 if dtype == np.float64 else 1e-5, cg_maxiter=500)
        tol = 1e-7 if dtype == np.float64 else 1e-3
    S, Wm = g['S'], g['W']
    S5 = S.reshape(S.shape[0], S.shape[1], 1, S.shape[2], 1)
    r = orc.admm_ccmod_maskdcpl(g['Z'], S5, Wm.reshape(S5.shape), tuple(int(v) for v in g['dsz']),
                                method=method, dtype=dtype, **kw)
    assert r['iters'] == int(g['k_final'])
    Nb = S.shape[2]
    # the reference keeps y0 on the filter axis of Y: (H, W, 1, 1, Nb + M)
    y0 = np.moveaxis(r['Y0'], 3, 4)
    u0 = np.moveaxis(r['U0'], 3, 4)
    assert rel_l2(y0, g['Y'][..., :Nb]) < tol and rel_l2(r['Y1'], g['Y'][..., Nb:]) < tol
    assert rel_l2(u0, g['U'][..., :Nb]) < tol and rel_l2(r['U1'], g['U'][..., Nb:]) < tol
    assert rel_l2(r['X'], g['X']) < tol and rel_l2(r['D'], g['D']) < tol
    for key in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key
    assert np.max(np.abs(r['Cnstr'] - g['it_Cnstr'])) < max(10 * tol, 1e-6)
    if method == 'cg':
        assert np.array_equal(r['XSlvCGIt'], g['it_XSlvCGIt'])


def test_pgm_mcdict_traces():
    """FISTA with a multi-channel dictionary: gradient summed over the channels
    (pgm/cbpdn.py:263-279)."""
    g = load_golden('pgm_mcdict_f64')
    D5, S5 = mcdict_inputs(g)
    r = orc.pgm_cbpdn(D5, S5, float(g['lmbda']), dtype=np.float64, maxiter=30, L=500.0,
                      rel_tol=0.0)
    assert r['iters'] == int(g['k_final'])
    assert r['X'].shape == g['X'].shape and rel_l2(r['X'], g['X']) < 1e-9
    for key in ('ObjFun', 'DFid', 'RegL1', 'Rsdl'):
        assert rel_l2(r[key], g['it_' + key]) < 1e-9, key


def test_admm_known_answer():
    g = load_golden('admm_known_answer_f64')
    D5, S5 = to5d(g['D'], g['S'])
    r = orc.admm_cbpdn(D5, S5, float(g['lmbda']), dtype=np.float64,
                       maxiter=500, rho=1e-1, auto_rho=False)
    assert r['iters'] == int(g['k_final'])
    assert rel_l2(r['Y'], g['Y']) < 1e-9
    # the reference's own assertion, tests/admm/test_cbpdn.py:173-176
    assert orc.rrs(g['X0'], r['Y'].squeeze()) < 5e-5


PGM_CASES = {
    'pgm_default_f64': dict(maxiter=40, L=500.0),
    'pgm_default_f32': dict(maxiter=40, L=500.0, dtype=np.float32),
    'pgm_nonneg_nobndry_f64': dict(maxiter=30, L=500.0, nonneg=True, nobndry=True),
    'pgm_multichan_f64': dict(maxiter=30, L=500.0),
}


@pytest.mark.parametrize('name', sorted(PGM_CASES))
def test_pgm_traces(name):
    g = load_golden(name)
    kw = dict(PGM_CASES[name])
    dtype = kw.pop('dtype', np.float64)
    tol = 1e-9 if dtype == np.float64 else 2e-4
    D5, S5 = to5d(g['D'], g['S'])
    r = orc.pgm_cbpdn(D5, S5, float(g['lmbda']), dtype=dtype, rel_tol=0.0, **kw)
    assert r['iters'] == int(g['k_final'])
    assert rel_l2(r['X'], g['X']) < tol
    for key in ('ObjFun', 'DFid', 'RegL1', 'Rsdl'):
        assert rel_l2(r[key], g['it_' + key]) < tol, key


def test_pcn():
    g = load_golden('pcn')
    for crp in (0, 1):
        for zm in (0, 1):
            y = orc.pcn(g['x'], (5, 5, 6), (16, 12), crp=bool(crp), zm=bool(zm))
            assert rel_l2(y, g['pcn_crp%d_zm%d' % (crp, zm)]) < 1e-14
