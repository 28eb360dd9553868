#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ from the UNMODIFIED reference.

TEST INFRASTRUCTURE ONLY.  Runs only in the authoring container, where the
reference is mounted read-only at /root/reference (it does not exist on the
GPU box, so nothing under tests/ -m gpu, smoke() or bench.py reads it).

    python oracle/make_golden.py            # (re)writes tests/golden/*.npz

The reference is pure Python; three of its import-time dependencies
(`future`, `imageio`, `filetype`) are absent from this image and are
replaced by the tiny stand-ins in oracle/_stubs.
pyFFTW is absent too, so the reference runs on its own numpy.fft fallback
(sporco/fft.py:593-639).

Each fixture stores the seeded inputs, the option values and the reference
outputs (final iterates plus per-iteration IterationStats traces).
"""

import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get('SPORCO_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, '_stubs'))
warnings.filterwarnings('ignore')

from sporco.admm import cbpdn as ref_cbpdn            # noqa: E402
from sporco.pgm import cbpdn as ref_pgm_cbpdn         # noqa: E402
from sporco.pgm.backtrack import BacktrackStandard, BacktrackRobust  # noqa: E402
from sporco.pgm.momentum import MomentumLinear, MomentumGenLinear    # noqa: E402
from sporco.pgm.stepsize import StepSizePolicyBB, StepSizePolicyCauchy  # noqa: E402
from sporco.dictlrn import cbpdndl as ref_cbpdndl     # noqa: E402
from sporco.pgm import ccmod as ref_pgm_ccmod         # noqa: E402
from sporco.admm import ccmod as ref_admm_ccmod       # noqa: E402
from sporco import linalg as ref_linalg               # noqa: E402
from sporco import prox as ref_prox                   # noqa: E402
from sporco import fft as ref_fft                     # noqa: E402
from sporco import cnvrep as ref_cnvrep               # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-34s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


def itstat_dict(b, prefix='it_'):
    its = b.getitstat()
    out = {}
    for f in its._fields:
        if f == 'Time':
            continue
        v = getattr(its, f)
        if len(v) and v[0] is None:
            continue
        out[prefix + f] = np.asarray(v, dtype=np.float64)
    return out


# ---------------------------------------------------------------------------
def admm_case(name, D, S, lmbda, optd, dimK=None, joint_mu=None, dimN=2):
    if joint_mu is None:
        opt = ref_cbpdn.ConvBPDN.Options(optd)
        b = ref_cbpdn.ConvBPDN(D, S, lmbda, opt, dimK=dimK, dimN=dimN)
    else:
        opt = ref_cbpdn.ConvBPDNJoint.Options(optd)
        b = ref_cbpdn.ConvBPDNJoint(D, S, lmbda, joint_mu, opt, dimK=dimK, dimN=dimN)
    b.solve()
    extra = {}
    for key, val in optd.items():
        if isinstance(val, np.ndarray):
            extra['optarr_' + key] = val.copy()
    save(name, D=D, S=S, lmbda=np.float64(lmbda),
         mu=np.float64(-1.0 if joint_mu is None else joint_mu),
         dimK=np.int64(-1 if dimK is None else dimK),
         X=b.X, Y=b.Y, U=b.U, Xf=b.Xf, rho_final=np.float64(b.rho),
         k_final=np.int64(b.k), recon=b.reconstruct(),
         **extra, **itstat_dict(b))


def gen_admm_cplx():
    """admm.cbpdn.ConvBPDN on complex-valued D and S (sporco/admm/cbpdn.py:209-217; the setting of
    the reference's tests/admm/test_cbpdn.py:179-201): default options (AutoRho, relaxation), two
    images, float64; a fixed-rho run with AuxVarObj; a complex64 run."""
    rng = np.random.RandomState(777)
    D = rng.randn(5, 5, 4) + 1j * rng.randn(5, 5, 4)
    S = rng.randn(16, 18, 2) + 1j * rng.randn(16, 18, 2)
    admm_case('admm_cplx_default_f64', D, S, 0.1, {'MaxMainIter': 30})
    admm_case('admm_cplx_fixedrho_auxvar_f64', D, S[..., 0], 0.05,
              {'MaxMainIter': 25, 'rho': 2.0, 'RelaxParam': 1.0, 'AutoRho': {'Enabled': False},
               'AuxVarObj': True})
    admm_case('admm_cplx_default_f32', D.astype(np.complex64), S.astype(np.complex64), 0.1,
              {'MaxMainIter': 30})


def gen_dim1():
    """dimN = 1: one-dimensional signals (sporco/cnvrep.py:33-198 with dimN=1; the constructor
    contract of sporco/admm/cbpdn.py:175, pgm/cbpdn.py:100).  A single signal, three signals
    (dimK = 1), three channels with the joint l2,1 term; FISTA on the three signals."""
    rng = np.random.RandomState(1001)
    D = rng.randn(6, 5)
    admm_case('admm_dim1_single_f64', D, rng.randn(48), 0.1, {'MaxMainIter': 30}, dimN=1)
    admm_case('admm_dim1_multi_f64', D, rng.randn(45, 3), 0.1, {'MaxMainIter': 30, 'NonNegCoef': True},
              dimK=1, dimN=1)
    admm_case('admm_dim1_joint_f64', D, rng.randn(40, 3), 0.05, {'MaxMainIter': 25}, dimK=0, joint_mu=0.02,
              dimN=1)
    S = rng.randn(45, 3)
    opt = ref_pgm_cbpdn.ConvBPDN.Options({'MaxMainIter': 30, 'L': 50.0, 'Backtrack': BacktrackStandard()})
    b = ref_pgm_cbpdn.ConvBPDN(D, S, 0.1, opt, dimK=1, dimN=1)
    X = b.solve()
    save('pgm_dim1_f64', D=D, S=S, lmbda=np.float64(0.1), X=X, k_final=np.int64(b.k), recon=b.reconstruct(),
         **itstat_dict(b))


def gen_dim1_dl():
    """dimN = 1 dictionary learning and the PGM dictionary update alone (sporco/dictlrn/cbpdndl.py:385,
    sporco/pgm/ccmod.py:139 with dimN=1): signals of three channels with a single-channel
    dictionary and zero-mean filters; a two-channel dictionary; the update with a mask."""
    rng = np.random.RandomState(1002)
    N, K, M, w = 48, 4, 5, 7
    for name, xm, S, D0, zm in (('cbpdndl_dim1_admm_f64', 'admm', rng.randn(N, 3, K), rng.randn(w, M), True),
                                ('cbpdndl_dim1_pgm_f64', 'pgm', rng.randn(N, K), rng.randn(w, M), False),
                                ('cbpdndl_dim1_mcdict_f64', 'admm', rng.randn(N, 2, K), rng.randn(w, 2, M), False)):
        opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
            {'MaxMainIter': 10, 'AccurateDFid': True, 'CCMOD': {'ZeroMean': zm}}, xmethod=xm, dmethod='pgm')
        b = ref_cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod=xm, dmethod='pgm', dimK=1, dimN=1)
        D1 = b.solve()
        save(name, D0=D0, S=S, lmbda=np.float64(0.1), zm=np.bool_(zm), D1=D1, X=b.getcoef(),
             recon=b.reconstruct(), **itstat_dict(b))
    S = rng.randn(N, K)
    Z = rng.randn(N, 1, K, M) * (rng.rand(N, 1, K, M) > 0.6)
    opt = ref_pgm_ccmod.ConvCnstrMOD.Options({'MaxMainIter': 25, 'L': 60.0, 'ZeroMean': True})
    c = ref_pgm_ccmod.ConvCnstrMOD(Z, S, (w, M), opt, dimK=1, dimN=1)
    c.solve()
    W = (rng.rand(N, 1, K, 1) > 0.3).astype(np.float64)
    opt = ref_pgm_ccmod.ConvCnstrMODMask.Options({'MaxMainIter': 25, 'L': 60.0})
    cm = ref_pgm_ccmod.ConvCnstrMODMask(Z, S, W, (w, M), opt, dimK=1, dimN=1)
    cm.solve()
    save('pgm_ccmod_dim1_f64', Z=Z, S=S, W=W, dsz=np.array((w, M)), D=c.getdict(), Xfull=c.X,
         recon=c.reconstruct(), D_mask=cm.getdict(), DFid_mask=np.array(cm.getitstat().DFid),
         **itstat_dict(c))


def gen_dim1_dstep():
    """dimN = 1 in the ADMM dictionary updates (sporco/admm/ccmod.py with one-dimensional signals)
    alone and as the D-step of ConvBPDNDictLearn."""
    rng = np.random.RandomState(1003)
    N, K, M, w = 48, 4, 5, 7
    S = rng.randn(N, K)
    Z = rng.randn(N, 1, K, M) * (rng.rand(N, 1, K, M) > 0.5)
    D0 = rng.randn(w, M)
    for meth, cls in (('ism', ref_admm_ccmod.ConvCnstrMOD_IterSM), ('cg', ref_admm_ccmod.ConvCnstrMOD_CG),
                      ('cns', ref_admm_ccmod.ConvCnstrMOD_Consensus)):
        optd = {'MaxMainIter': 15, 'ZeroMean': True, 'LinSolveCheck': True}
        if meth == 'cg':
            optd['CG'] = {'MaxIter': 500, 'StopTol': 1e-9}
        c = cls(Z, S, (w, M), cls.Options(optd), dimK=1, dimN=1)
        c.solve()
        optl = {'MaxMainIter': 8, 'AccurateDFid': True, 'CCMOD': {'ZeroMean': True}}
        if meth == 'cg':
            optl['CCMOD']['CG'] = {'MaxIter': 500, 'StopTol': 1e-9}
        b = ref_cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, ref_cbpdndl.ConvBPDNDictLearn.Options(
            optl, xmethod='admm', dmethod=meth), xmethod='admm', dmethod=meth, dimK=1, dimN=1)
        D1 = b.solve()
        save('ccmod_dim1_%s_f64' % meth, Z=Z, S=S, D0=D0, dsz=np.array((w, M)), D=c.getdict(), Y=c.Y, X=c.X,
             U=c.U, k_final=np.int64(c.k), dl_D1=D1, dl_X=b.getcoef(), dl_ObjFun=np.array(b.getitstat().ObjFun),
             **itstat_dict(c))


def gen_dim3():
    """dimN = 3: volumes (sporco/cnvrep.py:33-198 with three spatial axes; the reference's
    examples/scripts/cdl/cbpdndl_video.py:74 is the use).  A single volume, two volumes with
    NonNegCoef, three channels with the joint l2,1 term and a per-filter L1Weight; FISTA with
    backtracking on the two volumes."""
    rng = np.random.RandomState(1004)
    D = rng.randn(3, 4, 4, 4)
    admm_case('admm_dim3_single_f64', D, rng.randn(6, 12, 10), 0.1, {'MaxMainIter': 25}, dimN=3)
    admm_case('admm_dim3_multi_f64', D, rng.randn(6, 12, 10, 2), 0.1, {'MaxMainIter': 25, 'NonNegCoef': True},
              dimK=1, dimN=3)
    w = np.abs(rng.randn(1, 1, 1, 1, 1, 4)) + 0.5
    admm_case('admm_dim3_joint_f64', D, rng.randn(5, 9, 8, 3), 0.05, {'MaxMainIter': 20, 'L1Weight': w},
              dimK=0, joint_mu=0.02, dimN=3)
    S = rng.randn(6, 12, 10, 2)
    opt = ref_pgm_cbpdn.ConvBPDN.Options({'MaxMainIter': 25, 'L': 100.0, 'Backtrack': BacktrackStandard()})
    b = ref_pgm_cbpdn.ConvBPDN(D, S, 0.1, opt, dimK=1, dimN=3)
    X = b.solve()
    save('pgm_dim3_f64', D=D, S=S, lmbda=np.float64(0.1), X=X, k_final=np.int64(b.k), recon=b.reconstruct(),
         **itstat_dict(b))


def gen_dim3_dl():
    """dimN = 3 dictionary learning: the configuration of the reference's
    examples/scripts/cdl/cbpdndl_video.py:64-74 (ADMM sparse coding, consensus dictionary update, a
    single volume, AutoRho in both steps) at a small size, the PGM dictionary update on two volumes,
    and the two dictionary updates alone."""
    rng = np.random.RandomState(1005)
    D0 = rng.randn(3, 3, 2, 4)
    S1, S2 = rng.randn(8, 10, 6), rng.randn(8, 10, 6, 2)
    lmbda = 0.1
    opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 10, 'CBPDN': {'rho': 50.0 * lmbda, 'AutoRho': {'Enabled': True}},
         'CCMOD': {'rho': 1e2, 'AutoRho': {'Enabled': True}}}, dmethod='cns')
    b = ref_cbpdndl.ConvBPDNDictLearn(D0, S1, lmbda, opt, dimK=0, dimN=3)
    D1 = b.solve()
    save('cbpdndl_dim3_video_f64', D0=D0, S=S1, lmbda=np.float64(lmbda), D1=D1, X=b.getcoef(),
         recon=b.reconstruct(), **itstat_dict(b))
    opt = ref_cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 10, 'AccurateDFid': True}, xmethod='admm',
                                                dmethod='pgm')
    b = ref_cbpdndl.ConvBPDNDictLearn(D0, S2, lmbda, opt, xmethod='admm', dmethod='pgm', dimK=1, dimN=3)
    D1 = b.solve()
    save('cbpdndl_dim3_pgm_f64', D0=D0, S=S2, lmbda=np.float64(lmbda), D1=D1, X=b.getcoef(),
         recon=b.reconstruct(), **itstat_dict(b))
    Z = rng.randn(8, 10, 6, 1, 2, 4) * (rng.rand(8, 10, 6, 1, 2, 4) > 0.6)
    dsz = (3, 3, 2, 4)
    c = ref_pgm_ccmod.ConvCnstrMOD(Z, S2, dsz, ref_pgm_ccmod.ConvCnstrMOD.Options({'MaxMainIter': 25, 'L': 150.0}),
                                   dimK=1, dimN=3)
    c.solve()
    n = ref_admm_ccmod.ConvCnstrMOD_Consensus(Z, S2, dsz, ref_admm_ccmod.ConvCnstrMOD_Consensus.Options(
        {'MaxMainIter': 20, 'LinSolveCheck': True, 'rho': 3.0}), dimK=1, dimN=3)
    n.solve()
    save('ccmod_dim3_f64', Z=Z, S=S2, dsz=np.array(dsz), pgm_D=c.getdict(), pgm_X=c.X, pgm_recon=c.reconstruct(),
         pgm_DFid=np.array(c.getitstat().DFid), pgm_Rsdl=np.array(c.getitstat().Rsdl),
         cns_D=n.getdict(), cns_Y=n.Y, cns_X=n.X, cns_U=n.U, cns_DFid=np.array(n.getitstat().DFid),
         cns_PrimalRsdl=np.array(n.getitstat().PrimalRsdl), cns_DualRsdl=np.array(n.getitstat().DualRsdl))


def gen_admm():
    np.random.seed(12345)
    D = np.random.randn(5, 5, 4)
    S = np.random.randn(16, 16, 2)
    # default options (AutoRho on, relaxation 1.8), both precisions
    admm_case('admm_default_f64', D, S, 0.1, {'MaxMainIter': 30})
    admm_case('admm_default_f32', D, S, 0.1,
              {'MaxMainIter': 30, 'DataType': np.float32})
    # LinSolveCheck + fixed rho + no relaxation
    admm_case('admm_fixedrho_f64', D, S, 0.05,
              {'MaxMainIter': 25, 'rho': 2.0, 'RelaxParam': 1.0,
               'AutoRho': {'Enabled': False}, 'LinSolveCheck': True})
    # non-default AutoRho: period 3, no autoscaling, std residuals
    admm_case('admm_autorho_std_f64', D, S, 0.1,
              {'MaxMainIter': 30, 'AutoRho': {'Period': 3, 'AutoScaling': False,
                                             'Scaling': 2.0, 'RsdlRatio': 1.5,
                                             'StdResiduals': True},
               'AbsStopTol': 1e-6})
    # odd sizes, single image (dimK inferred 0), NonNeg + NoBndryCross
    S2 = np.random.randn(15, 17)
    admm_case('admm_odd_nonneg_nobndry_f64', D, S2, 0.1,
              {'MaxMainIter': 30, 'NonNegCoef': True, 'NoBndryCross': True})
    # L1Weight per filter + AuxVarObj (objective from Y)
    w = np.abs(np.random.randn(1, 1, 1, 1, 4)) + 0.5
    admm_case('admm_l1weight_auxvar_f64', D, S, 0.1,
              {'MaxMainIter': 25, 'L1Weight': w, 'AuxVarObj': True})
    # spatially varying L1Weight, already in internal 5-D layout (H, W, 1, N, 1)
    w2 = np.abs(np.random.randn(16, 16, 1, 2, 1)) + 0.5
    admm_case('admm_l1weight_spatial_f64', D, S, 0.1,
              {'MaxMainIter': 20, 'L1Weight': w2})
    # multi-channel signal, single-channel dictionary, multiple images
    S3 = np.random.randn(16, 12, 3, 2)
    admm_case('admm_multichan_f64', D, S3, 0.1, {'MaxMainIter': 25})
    # joint l2,1
    admm_case('admm_joint_f64', D, S3, 0.1, {'MaxMainIter': 25}, joint_mu=0.05)
    admm_case('admm_joint_f32', D, S3, 0.1,
              {'MaxMainIter': 25, 'DataType': np.float32}, joint_mu=0.05)
    # L21Weight must broadcast against both (H,W,1,N,K) (prox) and (H,W,N,K)
    # (objective, no keepdims) in the reference => trailing (N, K) shape
    wj = np.abs(np.random.randn(2, 4)) + 0.5
    admm_case('admm_joint_l21weight_f64', D, S3, 0.1,
              {'MaxMainIter': 20, 'L21Weight': wj, 'NonNegCoef': True},
              joint_mu=0.1)
    # warm start through Y0 and U0.  (Y0 alone raises AttributeError in the
    # reference: ConvBPDN.uinit, cbpdn.py:601-610, reads self.lmbda before
    # ConvBPDN.__init__ has set it.)
    Y0 = np.random.randn(16, 16, 1, 2, 4) * (np.random.rand(16, 16, 1, 2, 4) > 0.8)
    U0 = 0.1 * np.random.randn(16, 16, 1, 2, 4)
    admm_case('admm_warmstart_f64', D, S, 0.1,
              {'MaxMainIter': 15, 'Y0': Y0, 'U0': U0})
    # default lambda (cbpdn.py:573-578)
    b = ref_cbpdn.ConvBPDN(D, S, None, ref_cbpdn.ConvBPDN.Options({'MaxMainIter': 5}))
    b.solve()
    save('admm_default_lambda', D=D, S=S, lmbda=np.float64(b.lmbda),
         rho0=np.float64(50.0 * b.lmbda + 1.0), Y=b.Y, **itstat_dict(b))


def gen_known_answer():
    """Recipe of tests/admm/test_cbpdn.py:156-176 (sparse synthesis, fixed rho)."""
    np.random.seed(12345)
    N, M, Nd = 64, 4, 8
    D = np.random.randn(Nd, Nd, M)
    X0 = np.zeros((N, N, M))
    xr = np.random.randn(N, N, M)
    xp = np.abs(xr) > 3
    X0[xp] = np.random.randn(X0[xp].size)
    S = np.sum(ref_fft.fftconv(D, X0, axes=(0, 1)), axis=2)
    opt = ref_cbpdn.ConvBPDN.Options({'Verbose': False, 'MaxMainIter': 500,
                                      'RelStopTol': 1e-3, 'rho': 1e-1,
                                      'AutoRho': {'Enabled': False}})
    b = ref_cbpdn.ConvBPDN(D, S, 1e-4, opt)
    b.solve()
    save('admm_known_answer_f64', D=D, S=S, X0=X0, lmbda=np.float64(1e-4),
         Y=b.Y, k_final=np.int64(b.k), recon=b.reconstruct(), **itstat_dict(b))
    # odd size variant: tests/admm/test_cbpdn.py:204-225
    N = 63
    X0 = np.zeros((N, N, M))
    xr = np.random.randn(N, N, M)
    xp = np.abs(xr) > 3
    X0[xp] = np.random.randn(X0[xp].size)
    S = np.sum(np.fft.ifftn(np.fft.fftn(D, (N, N), (0, 1)) *
                            np.fft.fftn(X0, None, (0, 1)), None, (0, 1)).real,
               axis=2)
    b = ref_cbpdn.ConvBPDN(D, S, 1e-4, opt)
    b.solve()
    save('admm_known_answer_odd_f64', D=D, S=S, X0=X0, lmbda=np.float64(1e-4),
         Y=b.Y, k_final=np.int64(b.k), **itstat_dict(b))


def gen_config1():
    """BASELINE config 1 shape: 256x256, K=32 8x8 filters, N=1, float32,
    default options; stores traces, norms and a strided subsample of Y."""
    rng = np.random.RandomState(12345)
    D = rng.randn(8, 8, 32).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(256, 256).astype(np.float32)
    opt = ref_cbpdn.ConvBPDN.Options({'MaxMainIter': 20, 'RelStopTol': 0.0})
    b = ref_cbpdn.ConvBPDN(D, S, 0.05, opt, dimK=0)
    b.solve()
    Y = b.Y
    save('admm_config1_f32', seed=np.int64(12345), lmbda=np.float64(0.05),
         Y_sub=Y[::16, ::16].copy(), Y_l2=np.float64(np.linalg.norm(Y.astype(np.float64))),
         Y_l1=np.float64(np.abs(Y.astype(np.float64)).sum()),
         Y_nnz=np.int64(np.count_nonzero(Y)), **itstat_dict(b))
    # float64 run of the same problem: the accuracy yardstick for fp32 kernels
    opt = ref_cbpdn.ConvBPDN.Options({'MaxMainIter': 20, 'RelStopTol': 0.0,
                                      'DataType': np.float64})
    b = ref_cbpdn.ConvBPDN(D, S, 0.05, opt, dimK=0)
    b.solve()
    Y = b.Y
    save('admm_config1_f64', seed=np.int64(12345), lmbda=np.float64(0.05),
         Y_sub=Y[::16, ::16].copy(), Y_l2=np.float64(np.linalg.norm(Y)),
         Y_l1=np.float64(np.abs(Y).sum()), Y_nnz=np.int64(np.count_nonzero(Y)),
         **itstat_dict(b))


def _strided(Y):
    return Y[::16, ::16].copy()


def gen_config2():
    """BASELINE config 2 shape (512x512, K=64 8x8 filters, float32, default options) on
    N = 2 of the bench's own images (bench.make_problem, rank 0): per-iteration traces, a
    strided subsample of Y and its norms, from the float32 and the float64 reference run."""
    sys.path.insert(0, REPO)
    import bench
    D, S = bench.make_problem(512, 512, 64, 2, 0)
    for tag, dt, extra in (('f32', np.float32, {}), ('f64', np.float64, {'DataType': np.float64})):
        optd = {'MaxMainIter': 10, 'RelStopTol': 0.0}
        optd.update(extra)
        b = ref_cbpdn.ConvBPDN(D, S, 0.05, ref_cbpdn.ConvBPDN.Options(optd))
        b.solve()
        Y = b.Y
        save('admm_config2_n2_' + tag, lmbda=np.float64(0.05), Y_sub=_strided(Y),
             Y_l2=np.float64(np.linalg.norm(Y.astype(np.float64))),
             Y_l1=np.float64(np.abs(Y.astype(np.float64)).sum()),
             Y_nnz=np.int64(np.count_nonzero(Y)), **itstat_dict(b))


def gen_mixed_radix():
    """Image sizes of the mixed-radix register kernels (round 6: H, W in {320, 384, 448, 480}), the
    bench's own synthetic inputs (bench.make_problem, rank 0), default options, the float64 reference
    run of the float32 inputs: traces, a strided subsample of Y and its norms.
      admm_mr_384x384_k32_n2   the shape of the bench line's next_rows entry, two of its images
      admm_mr_480x320_k64_n1   ... and of the K = 64 entry (H = 30 x 16, W = 20 x 16)
      admm_mr_448x384_k8_nonneg_n2   the 28-point transform, NonNegCoef, AutoRho period 2
      admm_mr_240x320_k64_n2   the third next_rows shape: an odd number of points per thread (15)"""
    sys.path.insert(0, REPO)
    import bench
    for name, (H, W, K, N), extra in (
            ('admm_mr_384x384_k32_n2', (384, 384, 32, 2), {}),
            ('admm_mr_480x320_k64_n1', (480, 320, 64, 1), {}),
            ('admm_mr_448x384_k8_nonneg_n2', (448, 384, 8, 2), {'NonNegCoef': True, 'AutoRho': {'Period': 2}}),
            ('admm_mr_240x320_k64_n2', (240, 320, 64, 2), {})):
        D, S = bench.make_problem(H, W, K, N, 0)
        optd = {'MaxMainIter': 10, 'RelStopTol': 0.0, 'DataType': np.float64}
        optd.update(extra)
        b = ref_cbpdn.ConvBPDN(D, S, 0.05, ref_cbpdn.ConvBPDN.Options(optd))
        b.solve()
        Y = b.Y
        save(name, lmbda=np.float64(0.05), shape=np.array((H, W, K, N)), Y_sub=Y[::8, ::8].copy(),
             Y_l2=np.float64(np.linalg.norm(Y)), Y_l1=np.float64(np.abs(Y).sum()),
             Y_nnz=np.int64(np.count_nonzero(Y)), **itstat_dict(b))


def gen_config5():
    """BASELINE config 5 kernels end to end: ConvBPDNDictLearn (xmethod admm, dmethod pgm),
    256x256, K=64 8x8 filters, N=4 images, 4 outer iterations, from the float64 reference run
    of the float32 inputs (seeded: RandomState(515))."""
    rng = np.random.RandomState(515)
    D0 = rng.randn(8, 8, 64).astype(np.float32)
    S = rng.randn(256, 256, 4).astype(np.float32)
    opt = ref_cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 4, 'AccurateDFid': True},
                                                xmethod='admm', dmethod='pgm')
    b = ref_cbpdndl.ConvBPDNDictLearn(D0.astype(np.float64), S.astype(np.float64), 0.1, opt,
                                      xmethod='admm', dmethod='pgm')
    D1 = b.solve()
    X = b.getcoef()
    save('cbpdndl_config5_n4_f64', lmbda=np.float64(0.1), D1=D1, X_sub=_strided(X),
         X_l2=np.float64(np.linalg.norm(X)), **itstat_dict(b))


def gen_config3():
    """BASELINE config 3's kernels on ONE image of the bench's own input (bench.make_problem_rgb,
    rank 0): ConvBPDNJoint, 512x512 RGB, K=128 8x8 filters, lambda 0.1, mu 0.01, default options,
    6 iterations, from the float64 reference run of the float32 inputs (100 M elements per array:
    a strided subsample of Y, its norms and the traces are kept)."""
    sys.path.insert(0, REPO)
    import bench
    D, S = bench.make_problem_rgb(512, 512, 128, 1, 0)
    opt = ref_cbpdn.ConvBPDNJoint.Options({'MaxMainIter': 6, 'RelStopTol': 0.0,
                                           'DataType': np.float64})
    b = ref_cbpdn.ConvBPDNJoint(D, S, 0.1, 0.01, opt)
    b.solve()
    Y = b.Y
    save('admm_config3_joint_n1_f64', lmbda=np.float64(0.1), mu=np.float64(0.01),
         Y_sub=_strided(Y), Y_l2=np.float64(np.linalg.norm(Y)),
         Y_l1=np.float64(np.abs(Y).sum()), Y_nnz=np.int64(np.count_nonzero(Y)),
         **itstat_dict(b))


def gen_config4():
    """BASELINE config 4's kernels on N = 2 of the bench's own images (bench.make_problem,
    rank 0): pgm.cbpdn.ConvBPDN, 512x512, K=64, lambda 0.05, L = 500 (class default), 8
    iterations, with and without BacktrackStandard, float64 reference runs of the float32
    inputs: subsample of X, norms, traces."""
    sys.path.insert(0, REPO)
    import bench
    D, S = bench.make_problem(512, 512, 64, 2, 0)
    for tag, bt in (('', None), ('_bt', BacktrackStandard())):
        optd = {'MaxMainIter': 8, 'RelStopTol': 0.0, 'L': 500.0, 'DataType': np.float64}
        if bt is not None:
            optd['Backtrack'] = bt
        b = ref_pgm_cbpdn.ConvBPDN(D, S, 0.05, ref_pgm_cbpdn.ConvBPDN.Options(optd))
        X = b.solve()
        save('pgm_config4_n2%s_f64' % tag, lmbda=np.float64(0.05), X_sub=_strided(X),
             X_l2=np.float64(np.linalg.norm(X)), X_l1=np.float64(np.abs(X).sum()),
             X_nnz=np.int64(np.count_nonzero(X)), L_final=np.float64(b.L), **itstat_dict(b))


def gen_tol():
    """Time-to-tolerance known answer at the config 2 shape: the sparse-synthesis input of
    bench.make_structured_problem (512x512, K=64, N=2), lambda 0.01, default options,
    RelStopTol 1e-3: the iteration count at which the reference stops (a
    backend must stop within +-1 of it), its traces and a subsample of the final Y."""
    sys.path.insert(0, REPO)
    import bench
    which = os.environ.get('GOLDEN_TOL_CASES', 'f32,f64').split(',')
    D, S = bench.make_structured_problem(512, 512, 64, 2, 0)
    for tag, extra in (('f32', {}), ('f64', {'DataType': np.float64})):
        if tag not in which:
            continue
        optd = {'MaxMainIter': 1000, 'RelStopTol': 1e-3}
        optd.update(extra)
        b = ref_cbpdn.ConvBPDN(D, S, 0.01, ref_cbpdn.ConvBPDN.Options(optd))
        b.solve()
        Y = b.Y
        save('admm_tol_config2_n2_' + tag, lmbda=np.float64(0.01), k_final=np.int64(b.k),
             Y_sub=_strided(Y), Y_l2=np.float64(np.linalg.norm(Y.astype(np.float64))),
             **itstat_dict(b))


# ---------------------------------------------------------------------------
def pgm_case(name, D, S, lmbda, optd, tag_objs=None):
    optd = dict(optd)
    optd.setdefault('RelStopTol', 0.0)     # run all MaxMainIter iterations
    opt = ref_pgm_cbpdn.ConvBPDN.Options(optd)
    b = ref_pgm_cbpdn.ConvBPDN(D, S, lmbda, opt)
    X = b.solve()
    clean = {k: v for k, v in optd.items()
             if isinstance(v, (int, float, bool))}
    save(name, D=D, S=S, lmbda=np.float64(lmbda), X=X.copy(), Xf=b.Xf,
         L_final=np.float64(b.L), k_final=np.int64(b.k),
         recon=b.reconstruct(),
         **{'opt_' + k: np.float64(v) for k, v in clean.items()},
         **itstat_dict(b))


def gen_pgm():
    np.random.seed(12345)
    D = np.random.randn(5, 5, 4)
    S = np.random.randn(16, 16, 2)
    pgm_case('pgm_default_f64', D, S, 0.1, {'MaxMainIter': 40, 'L': 500.0})
    pgm_case('pgm_default_f32', D, S, 0.1,
             {'MaxMainIter': 40, 'L': 500.0, 'DataType': np.float32})
    pgm_case('pgm_nonneg_nobndry_f64', D, S, 0.1,
             {'MaxMainIter': 30, 'L': 500.0, 'NonNegCoef': True,
              'NoBndryCross': True})
    pgm_case('pgm_btstd_f64', D, S, 0.1,
             {'MaxMainIter': 30, 'L': 1.0, 'Backtrack': BacktrackStandard()})
    pgm_case('pgm_btrobust_f64', D, S, 0.1,
             {'MaxMainIter': 30, 'L': 1.0, 'Backtrack': BacktrackRobust()})
    pgm_case('pgm_momlinear_f64', D, S, 0.1,
             {'MaxMainIter': 30, 'L': 500.0, 'Momentum': MomentumLinear()})
    pgm_case('pgm_momgenlinear_f64', D, S, 0.1,
             {'MaxMainIter': 30, 'L': 500.0, 'Momentum': MomentumGenLinear()})
    pgm_case('pgm_stepbb_f64', D, S, 0.1,
             {'MaxMainIter': 30, 'L': 500.0, 'StepSizePolicy': StepSizePolicyBB()})
    pgm_case('pgm_stepcauchy_f64', D, S, 0.1,
             {'MaxMainIter': 30, 'L': 500.0,
              'StepSizePolicy': StepSizePolicyCauchy()})
    pgm_case('pgm_monotone_f64', D, S, 0.1,
             {'MaxMainIter': 30, 'L': 500.0, 'Monotone': True})
    S3 = np.random.randn(16, 12, 3, 2)
    pgm_case('pgm_multichan_f64', D, S3, 0.1, {'MaxMainIter': 30, 'L': 500.0})


def gen_pgm_btrobust256():
    """pgm.cbpdn.ConvBPDN with BacktrackRobust (sporco/pgm/backtrack.py:120-208) at a shape the
    fused FISTA kernels serve (the inputs of gen_pgm_bt256): float32 and float64 reference runs
    from L = 1 -- L grows by gamma_u over the first trials and shrinks by gamma_d every iteration."""
    g = np.load(os.path.join(OUT, 'pgm_bt256_f32.npz'))
    D, S = g['D'], g['S']
    for tag, extra in (('f32', {'DataType': np.float32}), ('f64', {'DataType': np.float64})):
        optd = {'MaxMainIter': 14, 'RelStopTol': 0.0, 'L': 1.0, 'Backtrack': BacktrackRobust()}
        optd.update(extra)
        b = ref_pgm_cbpdn.ConvBPDN(D, S, 0.02, ref_pgm_cbpdn.ConvBPDN.Options(optd))
        X = b.solve()
        save('pgm_btrobust256_' + tag, D=D, S=S, lmbda=np.float64(0.02), X_sub=_strided(X),
             X_l2=np.float64(np.linalg.norm(X.astype(np.float64))),
             X_nnz=np.int64(np.count_nonzero(X)), L_final=np.float64(b.L), **itstat_dict(b))


def gen_pgm_monotone256():
    """pgm.cbpdn.ConvBPDN with Monotone (sporco/pgm/pgm.py:804-811, :826-829) at a shape the fused
    FISTA kernels serve (the inputs of gen_pgm_bt256) and a step large enough (L = 8) that the
    objective goes up in three of the 14 iterations and the reference falls back."""
    g = np.load(os.path.join(OUT, 'pgm_bt256_f32.npz'))
    D, S = g['D'], g['S']
    for tag, extra in (('f32', {'DataType': np.float32}), ('f64', {'DataType': np.float64})):
        optd = {'MaxMainIter': 14, 'RelStopTol': 0.0, 'L': 8.0, 'Monotone': True}
        optd.update(extra)
        b = ref_pgm_cbpdn.ConvBPDN(D, S, 0.02, ref_pgm_cbpdn.ConvBPDN.Options(optd))
        X = b.solve()
        save('pgm_monotone256_' + tag, D=D, S=S, lmbda=np.float64(0.02), X_sub=_strided(X),
             X_l2=np.float64(np.linalg.norm(X.astype(np.float64))),
             Xf_l2=np.float64(np.linalg.norm(b.Xf.astype(np.complex128))),
             Yf_l2=np.float64(np.linalg.norm(b.Yf.astype(np.complex128))), **itstat_dict(b))


def gen_pgm_stepsize256():
    """pgm.cbpdn.ConvBPDN under StepSizePolicyCauchy and StepSizePolicyBB
    (sporco/pgm/stepsize.py:67-145) at a shape the fused FISTA kernels serve (the inputs of
    gen_pgm_bt256): float32 and float64 reference runs; L is the policy's from the third
    iteration on."""
    g = np.load(os.path.join(OUT, 'pgm_bt256_f32.npz'))
    D, S = g['D'], g['S']
    for pname, pol in (('cauchy', StepSizePolicyCauchy), ('bb', StepSizePolicyBB)):
        for tag, extra in (('f32', {'DataType': np.float32}), ('f64', {'DataType': np.float64})):
            optd = {'MaxMainIter': 14, 'RelStopTol': 0.0, 'L': 50.0, 'StepSizePolicy': pol()}
            optd.update(extra)
            b = ref_pgm_cbpdn.ConvBPDN(D, S, 0.02, ref_pgm_cbpdn.ConvBPDN.Options(optd))
            X = b.solve()
            save('pgm_step%s256_%s' % (pname, tag), D=D, S=S, lmbda=np.float64(0.02), X_sub=_strided(X),
                 X_l2=np.float64(np.linalg.norm(X.astype(np.float64))), L_final=np.float64(b.L),
                 **itstat_dict(b))


def gen_pgm_bt256():
    """pgm.cbpdn.ConvBPDN with BacktrackStandard at a shape the fused FISTA kernels serve
    (256x256, K = 8, 8x8 filters, one image), float32 and float64 reference runs from L = 1:
    the first iterations fail many trials, the later ones accept the first."""
    rng = np.random.RandomState(4242)
    D = rng.randn(8, 8, 8).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    X0 = np.zeros((256, 256, 8), np.float32)
    idx = rng.rand(*X0.shape) < 0.004
    X0[idx] = rng.randn(int(idx.sum())).astype(np.float32)
    S = np.sum(np.fft.irfft2(np.fft.rfft2(D, (256, 256), axes=(0, 1)) *
                             np.fft.rfft2(X0, axes=(0, 1)), (256, 256), axes=(0, 1)), axis=2)
    S = (S + 0.01 * rng.randn(256, 256)).astype(np.float32)
    for tag, extra in (('f32', {'DataType': np.float32}), ('f64', {'DataType': np.float64})):
        optd = {'MaxMainIter': 14, 'RelStopTol': 0.0, 'L': 1.0,
                'Backtrack': BacktrackStandard(gamma_u=1.5)}
        optd.update(extra)
        b = ref_pgm_cbpdn.ConvBPDN(D, S, 0.02, ref_pgm_cbpdn.ConvBPDN.Options(optd))
        X = b.solve()
        save('pgm_bt256_' + tag, D=D, S=S, lmbda=np.float64(0.02), X_sub=_strided(X),
             X_l2=np.float64(np.linalg.norm(X.astype(np.float64))),
             X_nnz=np.int64(np.count_nonzero(X)), L_final=np.float64(b.L), **itstat_dict(b))


# ---------------------------------------------------------------------------
def gen_primitives():
    np.random.seed(12345)
    # solvedbi_sm: shapes of tests/test_linalg.py:147-159
    N, M, K = 16, 4, 2
    ah = np.random.randn(N, N, 1, 1, M) + 1j * np.random.randn(N, N, 1, 1, M)
    b = np.random.randn(N, N, 1, K, M) + 1j * np.random.randn(N, N, 1, K, M)
    rho = 0.37
    x = ref_linalg.solvedbi_sm(ah, rho, b, axis=4)
    c = ref_linalg.solvedbi_sm_c(ah, np.conj(ah), rho, axis=4)
    ip = ref_linalg.inner(ah, b, axis=4)
    # real FFTs incl. odd sizes
    a = np.random.randn(12, 9, 3, 2, 4)
    af = ref_fft.rfftn(a, None, (0, 1))
    ar = ref_fft.irfftn(af, (12, 9), (0, 1))
    a2 = np.random.randn(16, 16, 1, 2, 4).astype(np.float32)
    a2f = ref_fft.rfftn(a2, None, (0, 1))
    # zero-padded dictionary transform (setdict, cbpdn.py:247)
    d = np.random.randn(5, 5, 1, 1, 4)
    df = ref_fft.rfftn(d, (12, 9), (0, 1))
    # Parseval
    nrm = ref_fft.rfl2norm2(af, a.shape, axis=(0, 1))
    nrm_even = ref_fft.rfl2norm2(a2f, a2.shape, axis=(0, 1))
    # prox
    v = np.random.randn(12, 9, 3, 2, 4)
    alpha = 0.3
    wgt = np.abs(np.random.randn(1, 1, 1, 1, 4))
    pl1 = ref_prox.prox_l1(v, alpha)
    pl1w = ref_prox.prox_l1(v, alpha * wgt)
    pl2 = ref_prox.prox_l2(v, 0.9, axis=2)
    psl = ref_prox.prox_sl1l2(v, alpha, 0.9, axis=2)
    vz = v.copy()
    vz[:, :, :, 0, 0] = 0.0     # exercises the zdivide branch
    pslz = ref_prox.prox_sl1l2(vz, alpha, 0.9, axis=2)
    save('primitives', sm_ah=ah, sm_b=b, sm_rho=np.float64(rho), sm_x=x,
         sm_c=c, inner_ab=ip, fft_a=a, fft_af=af, fft_ar=ar, fft_a2=a2,
         fft_a2f=a2f, fft_d=d, fft_df=df, nrm_odd=np.float64(nrm),
         nrm_even=np.float64(nrm_even), prox_v=v, prox_alpha=np.float64(alpha),
         prox_w=wgt, prox_l1=pl1, prox_l1w=pl1w, prox_l2=pl2, prox_sl1l2=psl,
         prox_vz=vz, prox_sl1l2_z=pslz)


def gen_pcn():
    np.random.seed(12345)
    x = np.random.randn(16, 12, 1, 1, 6)
    dsz = (5, 5, 6)
    out = {}
    for crp in (False, True):
        for zm in (False, True):
            y = ref_cnvrep.Pcn(x, dsz, (16, 12), dimN=2, dimC=1, crp=crp, zm=zm)
            out['pcn_crp%d_zm%d' % (crp, zm)] = y
    save('pcn', x=x, **out)


def gen_dictlearn():
    np.random.seed(12345)
    N, M, Nd, K = 16, 4, 5, 3
    D0 = np.random.randn(Nd, Nd, M)
    S = np.random.randn(N, N, K)
    lmbda = 0.1
    for name, dt in (('cbpdndl_f64', np.float64), ('cbpdndl_f32', np.float32)):
        opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
            {'MaxMainIter': 12, 'AccurateDFid': True},
            xmethod='admm', dmethod='pgm')
        b = ref_cbpdndl.ConvBPDNDictLearn(D0.astype(dt), S.astype(dt), lmbda,
                                          opt, xmethod='admm', dmethod='pgm')
        D1 = b.solve()
        save(name, D0=D0, S=S, lmbda=np.float64(lmbda), D1=D1,
             X=b.getcoef(), **itstat_dict(b))
    # ReturnX: the D-step is fed the X variable of the sparse coding step (dictlrn.py:379-382
    # through admm.py:955-956), not the default Y
    opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 8, 'AccurateDFid': True, 'CBPDN': {'ReturnX': True}},
        xmethod='admm', dmethod='pgm')
    b = ref_cbpdndl.ConvBPDNDictLearn(D0, S, lmbda, opt, xmethod='admm', dmethod='pgm')
    D1 = b.solve()
    save('cbpdndl_returnx_f64', D0=D0, S=S, lmbda=np.float64(lmbda), D1=D1, X=b.getcoef(),
         recon=b.reconstruct(), **itstat_dict(b))
    # PGM D-step alone with known coefficients (pgm/ccmod.py)
    X = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.7)
    opt = ref_pgm_ccmod.ConvCnstrMOD.Options({'MaxMainIter': 25, 'L': 800.0})
    c = ref_pgm_ccmod.ConvCnstrMOD(X, S, (Nd, Nd, M), opt)
    c.solve()
    save('pgm_ccmod_f64', Z=X, S=S, dsz=np.array((Nd, Nd, M)),
         D=c.getdict(), Xfull=c.X, **itstat_dict(c))


def gradreg_case(name, D, S, lmbda, mu, optd, dimK=None):
    opt = ref_cbpdn.ConvBPDNGradReg.Options(optd)
    b = ref_cbpdn.ConvBPDNGradReg(D, S, lmbda, mu, opt, dimK=dimK)
    b.solve()
    extra = {}
    for key, val in optd.items():
        if isinstance(val, np.ndarray):
            extra['optarr_' + key] = val.copy()
    save(name, D=D, S=S, lmbda=np.float64(lmbda), mu=np.float64(mu),
         dimK=np.int64(-1 if dimK is None else dimK),
         X=b.X, Y=b.Y, U=b.U, Xf=b.Xf, rho_final=np.float64(b.rho),
         k_final=np.int64(b.k), recon=b.reconstruct(), GHGf=b.GHGf,
         **extra, **itstat_dict(b))


def gen_gradreg():
    """ConvBPDNGradReg (sporco/admm/cbpdn.py:992-1214): SURVEY.md 8(f) rank 1."""
    np.random.seed(2468)
    D = np.random.randn(5, 5, 4)
    S = np.random.randn(16, 16, 2)
    gradreg_case('admm_gradreg_f64', D, S, 0.1, 0.2, {'MaxMainIter': 30})
    gradreg_case('admm_gradreg_f32', D, S, 0.1, 0.2,
                 {'MaxMainIter': 30, 'DataType': np.float32})
    # per-filter gradient weights (the usual use: penalise only a low-pass filter),
    # LinSolveCheck, fixed rho, odd size, single image
    wg = np.array([0.0, 0.0, 1.0, 0.5])
    S2 = np.random.randn(15, 17)
    gradreg_case('admm_gradreg_weights_f64', D, S2, 0.05, 0.5,
                 {'MaxMainIter': 25, 'GradWeight': wg, 'rho': 1.5,
                  'AutoRho': {'Enabled': False}, 'LinSolveCheck': True,
                  'NonNegCoef': True})
    # objective evaluated at the auxiliary variable: RegGrad then uses rfftn(Y)
    gradreg_case('admm_gradreg_auxvar_f64', D, S, 0.1, 0.3,
                 {'MaxMainIter': 20, 'AuxVarObj': True, 'GradWeight': wg})


def ams_case(name, cls, D, S, W, args, optd, dimK=None):
    opt = cls.Options(optd)
    b = ref_cbpdn.AddMaskSim(cls, D, S, W, *args, opt=opt, dimK=dimK)
    Xret = b.solve()
    c = b.cbpdn
    extra = {}
    for key, val in optd.items():
        if isinstance(val, np.ndarray):
            extra['optarr_' + key] = val.copy()
    save(name, D=D, S=S, W=W, lmbda=np.float64(args[0]),
         mu=np.float64(args[1] if len(args) > 1 else -1.0),
         dimK=np.int64(-1 if dimK is None else dimK),
         Wint=b.W, Xret=Xret, X=c.X, Y=c.Y, U=c.U, rho_final=np.float64(c.rho),
         k_final=np.int64(c.k), recon=b.reconstruct(), coef=b.getcoef(),
         **extra, **itstat_dict(c))


def gen_mcdict():
    """Multi-channel dictionary (Cd = C > 1): the solvemdbi_ism X-step branch,
    sporco/admm/cbpdn.py:277-279, sporco/linalg.py:370-444.  SURVEY.md 8(f) rank 2."""
    np.random.seed(8642)
    D = np.random.randn(5, 5, 3, 4)
    S = np.random.randn(16, 16, 3, 2)
    admm_case('admm_mcdict_f64', D, S, 0.1, {'MaxMainIter': 25, 'LinSolveCheck': True})
    admm_case('admm_mcdict_f32', D, S, 0.1, {'MaxMainIter': 25, 'DataType': np.float32})
    S1 = np.random.randn(15, 18, 3)
    admm_case('admm_mcdict_single_nonneg_f64', D, S1, 0.05,
              {'MaxMainIter': 20, 'NonNegCoef': True, 'AuxVarObj': True,
               'rho': 2.0, 'AutoRho': {'Enabled': False}})
    # the FISTA solver with the same kind of dictionary (pgm/cbpdn.py:263-286, sum over channels)
    pgm_case('pgm_mcdict_f64', D, S, 0.1, {'MaxMainIter': 30, 'L': 500.0})
    # dictionary update and dictionary learning with a multi-channel dictionary
    # (pgm/ccmod.py:139-404 with Cd > 1: channel-joint normalisation, cnvrep.py:868-913)
    N_, M_, Nd_, K_ = 16, 4, 5, 3
    Sc = np.random.randn(N_, N_, 3, K_)
    Zc = np.random.randn(N_, N_, 1, K_, M_) * (np.random.rand(N_, N_, 1, K_, M_) > 0.7)
    opt = ref_pgm_ccmod.ConvCnstrMOD.Options({'MaxMainIter': 20, 'L': 800.0, 'ZeroMean': True})
    c = ref_pgm_ccmod.ConvCnstrMOD(Zc, Sc, (Nd_, Nd_, 3, M_), opt)
    c.solve()
    save('pgm_ccmod_mcdict_f64', Z=Zc, S=Sc, dsz=np.array((Nd_, Nd_, 3, M_)), D=c.getdict(),
         Xfull=c.X, **itstat_dict(c))
    D0c = np.random.randn(Nd_, Nd_, 3, M_)
    for name, dt in (('cbpdndl_mcdict_f64', np.float64), ('cbpdndl_mcdict_f32', np.float32)):
        opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
            {'MaxMainIter': 10, 'AccurateDFid': True}, xmethod='admm', dmethod='pgm')
        b = ref_cbpdndl.ConvBPDNDictLearn(D0c.astype(dt), Sc.astype(dt), 0.1, opt,
                                          xmethod='admm', dmethod='pgm')
        D1 = b.solve()
        save(name, D0=D0c, S=Sc, lmbda=np.float64(0.1), D1=D1, X=b.getcoef(), **itstat_dict(b))
    g = {}
    # the primitive itself, on random data (4 channels, 6 filters)
    ah = np.random.randn(7, 5, 4, 1, 6) + 1j * np.random.randn(7, 5, 4, 1, 6)
    b = np.random.randn(7, 5, 1, 3, 6) + 1j * np.random.randn(7, 5, 1, 3, 6)
    x = ref_linalg.solvemdbi_ism(ah, 1.7, b.copy(), 4, 2)
    save('solvemdbi_ism', ah=ah, b=b, rho=np.float64(1.7), x=x)


def gen_mcdict_classes():
    """Multi-channel dictionaries under the other ADMM classes (own seed: gen_mcdict's
    fixtures stay as they are)."""
    np.random.seed(86420)
    D = np.random.randn(5, 5, 3, 4)
    S = np.random.randn(16, 16, 3, 2)
    # the other ADMM classes with such a dictionary: ConvBPDNGradReg (solvemdbi_ism with the
    # diagonal mu GHGf + rho, cbpdn.py:1181-1184), ConvBPDNJoint (no channel axis left in X),
    # AddMaskSim (one impulse filter per channel, the mask's channels on the filter axis,
    # cbpdn.py:2337-2364)
    wg = np.array([0.5, 0.0, 1.0, 2.0])
    gradreg_case('admm_gradreg_mcdict_f64', D, S, 0.1, 0.2,
                 {'MaxMainIter': 20, 'LinSolveCheck': True, 'GradWeight': wg})
    gradreg_case('admm_gradreg_mcdict_f32', D, S, 0.1, 0.2,
                 {'MaxMainIter': 20, 'DataType': np.float32})
    admm_case('admm_joint_mcdict_f64', D, S, 0.1, {'MaxMainIter': 20}, joint_mu=0.05)
    Wc = (np.random.rand(16, 16, 3, 2) > 0.3).astype(np.float64)
    W2 = (np.random.rand(16, 16) > 0.3).astype(np.float64)
    ams_case('ams_cbpdn_mcdict_f64', ref_cbpdn.ConvBPDN, D, S, Wc, (0.1,), {'MaxMainIter': 20})
    ams_case('ams_cbpdn_mcdict_bcast_f64', ref_cbpdn.ConvBPDN, D, S, W2, (0.1,),
             {'MaxMainIter': 20, 'NonNegCoef': True, 'AuxVarObj': True})
    ams_case('ams_gradreg_mcdict_f64', ref_cbpdn.ConvBPDNGradReg, D, S, Wc, (0.1, 0.2),
             {'MaxMainIter': 15})


def gen_cns():
    """ADMM consensus dictionary update ConvCnstrMOD_Consensus (sporco/admm/ccmod.py:605-908,
    sporco/admm/admm.py:1441-1707) alone and inside ConvBPDNDictLearn(dmethod='cns').
    SURVEY.md 8(f) rank 3."""
    np.random.seed(13579)
    N, M, Nd, K = 16, 4, 5, 3
    S = np.random.randn(N, N, K)
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.7)
    for name, optd in (
            ('ccmod_cns_f64', {'MaxMainIter': 20}),
            ('ccmod_cns_f32', {'MaxMainIter': 20, 'DataType': np.float32}),
            ('ccmod_cns_autorho_zm_f64',
             {'MaxMainIter': 25, 'ZeroMean': True, 'LinSolveCheck': True, 'rho': 2.0,
              'RelaxParam': 1.5,
              'AutoRho': {'Enabled': True, 'Period': 2, 'Scaling': 2.0, 'RsdlRatio': 1.2,
                          'AutoScaling': True, 'RsdlTarget': 1.0}})):
        opt = ref_admm_ccmod.ConvCnstrMOD_Consensus.Options(optd)
        c = ref_admm_ccmod.ConvCnstrMOD_Consensus(Z, S, (Nd, Nd, M), opt)
        c.solve()
        save(name, Z=Z, S=S, dsz=np.array((Nd, Nd, M)), D=c.getdict(), Y=c.Y, X=c.X, U=c.U,
             rho_final=np.float64(c.rho), k_final=np.int64(c.k), **itstat_dict(c))
    # warm start from a dictionary (the Y0 path used by dictionary learning, uinit :734-742)
    D0 = np.random.randn(Nd, Nd, M)
    Y0 = ref_cnvrep.zpad(ref_cnvrep.stdformD(
        ref_cnvrep.Pcn(D0, (Nd, Nd, M), (N, N), 2, 0, crp=True), 1, M, 2), (N, N))
    opt = ref_admm_ccmod.ConvCnstrMOD_Consensus.Options({'MaxMainIter': 10, 'Y0': Y0})
    c = ref_admm_ccmod.ConvCnstrMOD_Consensus(Z, S, (Nd, Nd, M), opt)
    c.solve()
    save('ccmod_cns_y0_f64', Z=Z, S=S, dsz=np.array((Nd, Nd, M)), Y0=Y0, D=c.getdict(),
         Y=c.Y, U=c.U, **itstat_dict(c))
    # dictionary learning with the consensus D-step
    for name, dt in (('cbpdndl_cns_f64', np.float64), ('cbpdndl_cns_f32', np.float32)):
        opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
            {'MaxMainIter': 10, 'AccurateDFid': True}, xmethod='admm', dmethod='cns')
        b = ref_cbpdndl.ConvBPDNDictLearn(D0.astype(dt), S.astype(dt), 0.1, opt,
                                          xmethod='admm', dmethod='cns')
        D1 = b.solve()
        save(name, D0=D0, S=S, lmbda=np.float64(0.1), D1=D1, X=b.getcoef(), **itstat_dict(b))


def gen_cns_options():
    """ConvCnstrMOD_Consensus with the options the reference's own tests switch on
    (tests/admm/test_ccmod.py:153-259: LinSolveCheck, a multi-channel signal with dimK = 0 and
    with several images) and with the objective evaluated at the blocks (AuxVarObj False:
    sporco/admm/ccmod.py:870-889, admm/admm.py:1632-1646)."""
    np.random.seed(24680)
    N, M, K, Nc, Nd = 16, 4, 2, 3, 5
    cls = ref_admm_ccmod.ConvCnstrMOD_Consensus

    def run(name, Z, S, dsz, optd, dimK=1):
        c = cls(Z, S, dsz, cls.Options(optd), dimK=dimK)
        c.solve()
        save(name, Z=Z, S=S, dsz=np.array(dsz), dimK=np.int64(dimK), D=c.getdict(), Y=c.Y, U=c.U,
             X=c.X, rho_final=np.float64(c.rho), k_final=np.int64(c.k), **itstat_dict(c))
    Z = np.random.randn(N, N, 1, K + 1, M) * (np.random.rand(N, N, 1, K + 1, M) > 0.6)
    S = np.random.randn(N, N, K + 1)
    run('ccmod_cns_auxfalse_chk_zm_f64', Z, S, (Nd, Nd, M),
        {'MaxMainIter': 12, 'AuxVarObj': False, 'LinSolveCheck': True, 'ZeroMean': True})
    run('ccmod_cns_fevalx_f32', Z, S, (Nd, Nd, M),
        {'MaxMainIter': 12, 'fEvalX': True, 'DataType': np.float32})
    Zc = np.random.randn(N, N, Nc, 1, M) * (np.random.rand(N, N, Nc, 1, M) > 0.6)
    Sc = np.random.randn(N, N, Nc)
    run('ccmod_cns_chk_multichan_dimk0_f64', Zc, Sc, (Nd, Nd, 1, M),
        {'MaxMainIter': 12, 'LinSolveCheck': True}, dimK=0)
    Zck = np.random.randn(N, N, Nc, K, M) * (np.random.rand(N, N, Nc, K, M) > 0.6)
    Sck = np.random.randn(N, N, Nc, K)
    run('ccmod_cns_chk_multichan_f64', Zck, Sck, (Nd, Nd, 1, M),
        {'MaxMainIter': 12, 'LinSolveCheck': True})
    # the mask-decoupled consensus update with LinSolveCheck (tests/admm/test_ccmodmd.py:258-413)
    from sporco.admm import ccmodmd as ref_ccmodmd
    Wm = (np.random.rand(N, N, Nc, K) > 0.3).astype(np.float64)
    mcls = ref_ccmodmd.ConvCnstrMODMaskDcpl_Consensus
    c = mcls(Zck, Sck, Wm, (Nd, Nd, 1, M), mcls.Options({'MaxMainIter': 12, 'LinSolveCheck': True}))
    c.solve()
    save('ccmodmd_cns_chk_multichan_f64', Z=Zck, S=Sck, W=Wm, dsz=np.array((Nd, Nd, 1, M)),
         D=c.getdict(), Y=c.Y, U=c.U, X=c.X, rho_final=np.float64(c.rho), k_final=np.int64(c.k),
         **itstat_dict(c))


def gen_cns_mcdict():
    """ConvCnstrMOD_Consensus and ConvBPDNDictLearn(dmethod='cns') with a multi-channel (colour)
    dictionary (sporco/admm/ccmod.py:696-698, :766-822: one block per image, the channels
    share the image's system matrix; the reference's examples/scripts/cdl/cbpdndl_cns_clr.py)."""
    np.random.seed(13531)
    N, M, K, Nc, Nd = 16, 4, 3, 3, 5
    cls = ref_admm_ccmod.ConvCnstrMOD_Consensus
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.6)
    S = np.random.randn(N, N, Nc, K)
    for name, optd in (('ccmod_cns_mcdict_f64', {'MaxMainIter': 12}),
                       ('ccmod_cns_mcdict_opts_f64', {'MaxMainIter': 12, 'LinSolveCheck': True,
                                                      'ZeroMean': True, 'AuxVarObj': False}),
                       ('ccmod_cns_mcdict_f32', {'MaxMainIter': 12, 'DataType': np.float32})):
        c = cls(Z, S, (Nd, Nd, Nc, M), cls.Options(optd))
        c.solve()
        save(name, Z=Z, S=S, dsz=np.array((Nd, Nd, Nc, M)), D=c.getdict(), Y=c.Y, U=c.U, X=c.X,
             rho_final=np.float64(c.rho), k_final=np.int64(c.k), **itstat_dict(c))
    D0 = np.random.randn(Nd, Nd, Nc, M)
    opt = ref_cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                                xmethod='admm', dmethod='cns')
    b = ref_cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod='cns')
    D1 = b.solve()
    save('cbpdndl_cns_mcdict_f64', D0=D0, S=S, lmbda=np.float64(0.1), D1=D1, X=b.getcoef(),
         **itstat_dict(b))


def gen_ccmod_eq():
    """Single-copy ADMM dictionary updates ConvCnstrMOD_IterSM and ConvCnstrMOD_CG
    (sporco/admm/ccmod.py:433-601 on ConvCnstrMODBase :103-429) alone and inside
    ConvBPDNDictLearn(dmethod='ism' / 'cg').  SURVEY.md 8(f) rank 3."""
    np.random.seed(86420)
    N, M, Nd, K = 16, 4, 5, 3
    S = np.random.randn(N, N, K)
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.7)
    D0 = np.random.randn(Nd, Nd, M)
    Y0 = ref_cnvrep.zpad(ref_cnvrep.stdformD(
        ref_cnvrep.Pcn(D0, (Nd, Nd, M), (N, N), 2, 0, crp=True), 1, M, 2), (N, N))
    fixed = {'rho': 5.0, 'AutoRho': {'Enabled': False}}
    for meth, cls in (('ism', ref_admm_ccmod.ConvCnstrMOD_IterSM),
                      ('cg', ref_admm_ccmod.ConvCnstrMOD_CG)):
        for name, optd in (
                ('f64', {'MaxMainIter': 20}),
                ('f32', {'MaxMainIter': 20, 'DataType': np.float32}),
                ('fixedrho_zm_chk_f64', dict(fixed, MaxMainIter=20, ZeroMean=True,
                                             LinSolveCheck=True, RelaxParam=1.5)),
                ('auxobj_y0_f64', {'MaxMainIter': 12, 'AuxVarObj': True, 'Y0': Y0,
                                   'AutoRho': {'Period': 3, 'Scaling': 2.0,
                                               'AutoScaling': False, 'RsdlRatio': 1.5}})):
            if meth == 'cg' and 'fixedrho' in name:
                optd = dict(optd, CG={'MaxIter': 500, 'StopTol': 1e-9})
            opt = cls.Options(optd)
            c = cls(Z, S, (Nd, Nd, M), opt)
            c.solve()
            extra = {'Y0': Y0} if 'Y0' in optd else {}
            save('ccmod_%s_%s' % (meth, name), Z=Z, S=S, dsz=np.array((Nd, Nd, M)),
                 D=c.getdict(), Y=c.Y, X=c.X, U=c.U, rho_final=np.float64(c.rho),
                 k_final=np.int64(c.k), **extra, **itstat_dict(c))
        for dt, tag in ((np.float64, 'f64'), (np.float32, 'f32')):
            opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
                {'MaxMainIter': 10, 'AccurateDFid': True}, xmethod='admm', dmethod=meth)
            b = ref_cbpdndl.ConvBPDNDictLearn(D0.astype(dt), S.astype(dt), 0.1, opt,
                                              xmethod='admm', dmethod=meth)
            D1 = b.solve()
            save('cbpdndl_%s_%s' % (meth, tag), D0=D0, S=S, lmbda=np.float64(0.1), D1=D1,
                 X=b.getcoef(), **itstat_dict(b))


def gen_ccmod_cplx():
    """Complex-valued coefficient maps, signals and dictionary in the ADMM dictionary updates
    (sporco/admm/ccmod.py:219-231: fftn / ifftn in place of rfftn / irfftn; the reference's own
    tests/admm/test_ccmod.py:49-140): IterSM with the default options (AutoRho) and with
    LinSolveCheck / ZeroMean / a start dictionary, in complex64 too; CG run to 1e-9; consensus."""
    rng = np.random.RandomState(97531)
    N, M, Nd, K = 16, 4, 5, 3
    cn = lambda *shp: rng.randn(*shp) + 1j * rng.randn(*shp)
    S = cn(N, N, K)
    Z = cn(N, N, 1, K, M) * (rng.rand(N, N, 1, K, M) > 0.7)
    D0 = cn(Nd, Nd, M)
    Y0 = ref_cnvrep.zpad(ref_cnvrep.stdformD(
        ref_cnvrep.Pcn(D0, (Nd, Nd, M), (N, N), 2, 0, crp=True), 1, M, 2), (N, N))
    fixed = {'rho': 2.0, 'AutoRho': {'Enabled': False}}
    cases = (('ism', 'f64', ref_admm_ccmod.ConvCnstrMOD_IterSM, {'MaxMainIter': 20}),
             ('ism', 'f32', ref_admm_ccmod.ConvCnstrMOD_IterSM, {'MaxMainIter': 20, 'DataType': np.complex64}),
             ('ism', 'chk_zm_y0_f64', ref_admm_ccmod.ConvCnstrMOD_IterSM,
              {'MaxMainIter': 15, 'LinSolveCheck': True, 'ZeroMean': True, 'RelaxParam': 1.5, 'Y0': Y0,
               'AuxVarObj': True}),
             ('cg', 'tight_f64', ref_admm_ccmod.ConvCnstrMOD_CG,
              dict(fixed, MaxMainIter=15, LinSolveCheck=True, CG={'MaxIter': 500, 'StopTol': 1e-9})),
             ('cg', 'f64', ref_admm_ccmod.ConvCnstrMOD_CG, {'MaxMainIter': 15}),
             ('cns', 'f64', ref_admm_ccmod.ConvCnstrMOD_Consensus, {'MaxMainIter': 20}),
             ('cns', 'chk_zm_autorho_f64', ref_admm_ccmod.ConvCnstrMOD_Consensus,
              {'MaxMainIter': 15, 'LinSolveCheck': True, 'ZeroMean': True, 'AutoRho': {'Enabled': True},
               'Y0': Y0}))
    for meth, name, cls, optd in cases:
        c = cls(Z, S, (Nd, Nd, M), cls.Options(optd))
        c.solve()
        save('ccmod_cplx_%s_%s' % (meth, name), Z=Z, S=S, Y0=Y0, dsz=np.array((Nd, Nd, M)), D=c.getdict(),
             Y=c.Y, X=c.X, U=c.U, rho_final=np.float64(c.rho), k_final=np.int64(c.k), **itstat_dict(c))
    # one image (dimK = 0), odd sizes
    S1, Z1 = cn(15, 13), cn(15, 13, 1, 1, M) * (rng.rand(15, 13, 1, 1, M) > 0.6)
    c = ref_admm_ccmod.ConvCnstrMOD_IterSM(Z1, S1, (4, 6, M), ref_admm_ccmod.ConvCnstrMOD_IterSM.Options(
        {'MaxMainIter': 15}), dimK=0)
    c.solve()
    save('ccmod_cplx_ism_odd_single_f64', Z=Z1, S=S1, dsz=np.array((4, 6, M)), D=c.getdict(), Y=c.Y, X=c.X,
         U=c.U, rho_final=np.float64(c.rho), k_final=np.int64(c.k), **itstat_dict(c))


def gen_ccmod_eq_mcdict():
    """The single-copy ADMM dictionary updates with a MULTI-CHANNEL dictionary (Cd = C > 1):
    ConvCnstrMOD_IterSM / _CG (sporco/admm/ccmod.py:433-601: linalg.solvemdbi_ism / _cg with the
    channel axis of b broadcast against a channel-less Zf) and their mask-decoupled forms
    (sporco/admm/ccmodmd.py:573-760), with channel-less coefficient maps (H, W, 1, K, M) -- one
    matrix per frequency shared by the channels -- and with maps that carry the dictionary's
    channels (H, W, C, K, M), which the reference's broadcasting turns into C independent updates
    sharing rho, the projection and the residuals (its own tests/admm/test_ccmodmd.py:176-194).
    Also inside ConvBPDNDictLearn(dmethod='ism' / 'cg') on an RGB dictionary."""
    from sporco.admm import ccmodmd as ref_ccmodmd
    np.random.seed(24680)
    N, M, Nd, K, C = 16, 4, 5, 3, 3
    S = np.random.randn(N, N, C, K)
    Wm = (np.random.rand(N, N, C, K) > 0.3).astype(np.float64)
    Z1 = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.7)
    Zc = np.random.randn(N, N, C, K, M) * (np.random.rand(N, N, C, K, M) > 0.7)
    dsz = (Nd, Nd, C, M)
    tight = {'MaxIter': 500, 'StopTol': 1e-9}
    for meth, cls, mcls in (('ism', ref_admm_ccmod.ConvCnstrMOD_IterSM, ref_ccmodmd.ConvCnstrMODMaskDcpl_IterSM),
                            ('cg', ref_admm_ccmod.ConvCnstrMOD_CG, ref_ccmodmd.ConvCnstrMODMaskDcpl_CG)):
        for ztag, Z in (('', Z1), ('_zchan', Zc)):
            for name, optd in (
                    ('f64', {'MaxMainIter': 12}),
                    ('chk_zm_f64', {'MaxMainIter': 12, 'ZeroMean': True, 'LinSolveCheck': True,
                                    'RelaxParam': 1.5, 'AuxVarObj': True}),
                    ('f32', {'MaxMainIter': 12, 'DataType': np.float32})):
                if ztag and name != 'f64':
                    continue
                if meth == 'cg':
                    optd = dict(optd, CG=tight)
                c = cls(Z, S, dsz, cls.Options(optd))
                c.solve()
                save('ccmod_%s_mcdict%s_%s' % (meth, ztag, name), Z=Z, S=S, dsz=np.array(dsz),
                     D=c.getdict(), Y=c.Y, X=c.X, U=c.U, rho_final=np.float64(c.rho),
                     k_final=np.int64(c.k), **itstat_dict(c))
            # mask decoupling
            for name, optd in (('f64', {'MaxMainIter': 12}),
                               ('chk_f64', {'MaxMainIter': 12, 'LinSolveCheck': True,
                                            'AutoRho': {'Enabled': True}, 'ZeroMean': True})):
                if ztag and name != 'f64':
                    continue
                if meth == 'cg':
                    optd = dict(optd, CG=tight)
                c = mcls(Z, S, Wm, dsz, mcls.Options(optd))
                c.solve()
                save('ccmodmd_%s_mcdict%s_%s' % (meth, ztag, name), Z=Z, S=S, W=Wm,
                     dsz=np.array(dsz), D=c.getdict(), Y=c.Y, X=c.X, U=c.U,
                     rho_final=np.float64(c.rho), k_final=np.int64(c.k), **itstat_dict(c))
    # dictionary learning on an RGB dictionary with both updates
    D0 = np.random.randn(Nd, Nd, C, M)
    for meth in ('ism', 'cg'):
        optd = {'MaxMainIter': 8, 'AccurateDFid': True}
        if meth == 'cg':
            optd['CCMOD'] = {'CG': tight}
        opt = ref_cbpdndl.ConvBPDNDictLearn.Options(optd, xmethod='admm', dmethod=meth)
        b = ref_cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod=meth)
        D1 = b.solve()
        save('cbpdndl_%s_mcdict_f64' % meth, D0=D0, S=S, lmbda=np.float64(0.1), D1=D1,
             X=b.getcoef(), **itstat_dict(b))


def gen_ccmod_ism_many():
    """ConvCnstrMOD_IterSM over MORE than 8 images (and images x channels): the reference takes
    any number (sporco/admm/ccmod.py:433-604, linalg.solvemdbi_ism over axisK); also inside the
    masked dictionary update ConvCnstrMODMaskDcpl_IterSM (ccmodmd.py:573-654)."""
    np.random.seed(97531)
    N, M, Nd = 12, 3, 4
    for K, C, tag in ((11, 1, 'k11'), (5, 2, 'k5c2')):
        shpS = (N, N, K) if C == 1 else (N, N, C, K)
        S = np.random.randn(*shpS)
        Z = np.random.randn(N, N, C, K, M) * (np.random.rand(N, N, C, K, M) > 0.6)
        for name, optd in (('f64', {'MaxMainIter': 10}),
                           ('f32', {'MaxMainIter': 10, 'DataType': np.float32})):
            if C > 1 and name == 'f32':
                continue
            opt = ref_admm_ccmod.ConvCnstrMOD_IterSM.Options(optd)
            c = ref_admm_ccmod.ConvCnstrMOD_IterSM(Z, S, (Nd, Nd, M), opt)
            c.solve()
            save('ccmod_ism_%s_%s' % (tag, name), Z=Z, S=S, dsz=np.array((Nd, Nd, M)),
                 D=c.getdict(), Y=c.Y, X=c.X, U=c.U, rho_final=np.float64(c.rho),
                 k_final=np.int64(c.k), **itstat_dict(c))
    # masked
    K = 10
    S = np.random.randn(N, N, K)
    W = (np.random.rand(N, N, K) > 0.3).astype(np.float64)
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.6)
    from sporco.admm import ccmodmd as ref_ccmodmd
    opt = ref_ccmodmd.ConvCnstrMODMaskDcpl_IterSM.Options({'MaxMainIter': 10})
    c = ref_ccmodmd.ConvCnstrMODMaskDcpl_IterSM(Z, S, W, (Nd, Nd, M), opt)
    c.solve()
    save('ccmodmd_ism_k10_f64', Z=Z, S=S, W=W, dsz=np.array((Nd, Nd, M)), D=c.getdict(), Y=c.Y,
         X=c.X, U=c.U, rho_final=np.float64(c.rho), k_final=np.int64(c.k), **itstat_dict(c))


def gen_online():
    """Online dictionary learning dictlrn.onlinecdl.OnlineConvBPDNDictLearn
    (sporco/dictlrn/onlinecdl.py:33-460): one solve() per training image.  SURVEY.md 8(f)
    rank 4."""
    from sporco.dictlrn import onlinecdl as ref_online
    np.random.seed(75319)
    N, M, Nd = 16, 4, 5
    D0 = np.random.randn(Nd, Nd, M)
    imgs = np.random.randn(N, N, 5)
    for tag, dt in (('f64', np.float64), ('f32', np.float32)):
        opt = ref_online.OnlineConvBPDNDictLearn.Options(
            {'eta_a': 8.0, 'eta_b': 4.0, 'ZeroMean': tag == 'f64', 'DataType': dt,
             'CBPDN': {'MaxMainIter': 30}})
        b = ref_online.OnlineConvBPDNDictLearn(D0, 0.1, opt, dimK=0)
        Ds = []
        for i in range(imgs.shape[-1]):
            Ds.append(b.solve(imgs[..., i].astype(dt)).copy())
        save('onlinecdl_' + tag, D0=D0, S=imgs, lmbda=np.float64(0.1), Ds=np.stack(Ds),
             **itstat_dict(b))
    # mini-batches of two images (a multi-channel signal with a single-channel dictionary
    # fails inside the reference's own dstep: Sf keeps its channel axis, onlinecdl.py:316)
    Sc = np.random.randn(N, N, 2, 3)
    opt = ref_online.OnlineConvBPDNDictLearn.Options({'CBPDN': {'MaxMainIter': 20}})
    b = ref_online.OnlineConvBPDNDictLearn(D0, 0.1, opt, dimK=1)
    Ds = [b.solve(Sc[..., i]).copy() for i in range(Sc.shape[-1])]
    save('onlinecdl_batch_f64', D0=D0, S=Sc, lmbda=np.float64(0.1), Ds=np.stack(Ds),
         **itstat_dict(b))


def gen_shard():
    """Dictionary learning on four images in ONE process: what the two-rank image-sharded run
    of tests/test_dist_gloo.py must reproduce."""
    np.random.seed(4242)
    N, M, Nd, K = 16, 4, 5, 4
    D0 = np.random.randn(Nd, Nd, M)
    S = np.random.randn(N, N, K)
    opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 10, 'AccurateDFid': True, 'CCMOD': {'ZeroMean': True}},
        xmethod='admm', dmethod='pgm')
    b = ref_cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod='pgm')
    D1 = b.solve()
    save('cbpdndl_shard_f64', D0=D0, S=S, lmbda=np.float64(0.1), D1=D1, X=b.getcoef(),
         **itstat_dict(b))
    # ... and masked learning, with the mask-decoupling and the FISTA X-step
    from sporco.dictlrn import cbpdndlmd as ref_md
    Wd = (np.random.rand(N, N, 1, K) > 0.3).astype(np.float64)
    for xm in ('admm', 'pgm'):
        opt = ref_md.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                                   xmethod=xm, dmethod='pgm')
        b = ref_md.ConvBPDNMaskDictLearn(D0, S, 0.1, Wd, opt, xmethod=xm, dmethod='pgm')
        D1 = b.solve()
        save('cbpdndlmd_shard_%s_f64' % xm, D0=D0, S=S, W=Wd, lmbda=np.float64(0.1), D1=D1,
             X=b.getcoef(), **itstat_dict(b))


def gen_shard_cns():
    """The consensus dictionary updates on FOUR images in one process: what their two-rank
    image-sharded runs (the consensus average as an all-reduce) must reproduce."""
    from sporco.admm import ccmodmd as ref_ccmodmd
    from sporco.dictlrn import cbpdndlmd as ref_md
    np.random.seed(2424)
    N, M, Nd, K = 16, 4, 5, 4
    D0 = np.random.randn(Nd, Nd, M)
    S = np.random.randn(N, N, K)
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.7)
    Wd = (np.random.rand(N, N, 1, K) > 0.3).astype(np.float64)
    autorho = {'Enabled': True, 'Period': 3, 'Scaling': 2.0, 'RsdlRatio': 1.2,
               'AutoScaling': True, 'RsdlTarget': 1.0}
    optd = {'MaxMainIter': 15, 'ZeroMean': True, 'AutoRho': autorho}
    c = ref_admm_ccmod.ConvCnstrMOD_Consensus(Z, S, (Nd, Nd, M),
                                               ref_admm_ccmod.ConvCnstrMOD_Consensus.Options(optd))
    c.solve()
    save('ccmod_cns_shard_f64', Z=Z, S=S, dsz=np.array((Nd, Nd, M)), D=c.getdict(), Y=c.Y,
         k_final=np.int64(c.k), **itstat_dict(c))
    cm = ref_ccmodmd.ConvCnstrMODMaskDcpl_Consensus(
        Z, S, Wd, (Nd, Nd, M), ref_admm_ccmod.ConvCnstrMOD_Consensus.Options(optd))
    cm.solve()
    save('ccmodmd_cns_shard_f64', Z=Z, S=S, W=Wd, dsz=np.array((Nd, Nd, M)), D=cm.getdict(),
         Y=cm.Y, k_final=np.int64(cm.k), **itstat_dict(cm))
    opt = ref_cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                                xmethod='admm', dmethod='cns')
    b = ref_cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod='cns')
    D1 = b.solve()
    save('cbpdndl_shard_cns_f64', D0=D0, S=S, lmbda=np.float64(0.1), D1=D1, X=b.getcoef(),
         **itstat_dict(b))
    opt = ref_md.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                               xmethod='admm', dmethod='cns')
    b = ref_md.ConvBPDNMaskDictLearn(D0, S, 0.1, Wd, opt, xmethod='admm', dmethod='cns')
    D1 = b.solve()
    save('cbpdndlmd_shard_cns_f64', D0=D0, S=S, W=Wd, lmbda=np.float64(0.1), D1=D1,
         X=b.getcoef(), **itstat_dict(b))


def gen_maskdcpl():
    """admm.cbpdn.ConvBPDNMaskDcpl (sporco/admm/cbpdn.py:2066-2283): mask decoupling, the ADMM
    X-step of masked dictionary learning.  SURVEY.md 8(f) rank 3."""
    np.random.seed(31415)
    D = np.random.randn(5, 5, 4)
    S = np.random.randn(16, 16, 2)
    W = (np.random.rand(16, 16, 2) > 0.3).astype(np.float64)
    Wb = (np.random.rand(16, 16) > 0.3).astype(np.float64)       # one mask for all images
    wl1 = 0.5 + np.random.rand(1, 1, 1, 1, 4)
    Sc = np.random.randn(16, 16, 3, 2)
    Wc = (np.random.rand(16, 16, 3, 2) > 0.3).astype(np.float64)
    # multi-channel dictionaries (Cd > 1: X-step by solvemdbi_ism with rho = 1, block 0 swapped
    # onto the filter axis; the cases of the reference's tests/admm/test_cbpdn.py:487-518)
    Dm = np.random.randn(5, 5, 3, 4)
    S1 = np.random.randn(16, 16, 3)
    W1 = (np.random.rand(16, 16, 3) > 0.3).astype(np.float64)
    for name, DD, SS, WW, optd in (
            ('maskdcpl_mcdict_f64', Dm, Sc, Wc, {'MaxMainIter': 20}),
            ('maskdcpl_mcdict_one_image_opts_f64', Dm, S1, W1,
             {'MaxMainIter': 20, 'rho': 1.5, 'RelaxParam': 1.6, 'NonNegCoef': True,
              'AuxVarObj': True, 'LinSolveCheck': True,
              'AutoRho': {'Enabled': True, 'Period': 2, 'Scaling': 2.0, 'RsdlRatio': 1.2,
                          'AutoScaling': True, 'RsdlTarget': 1.0}}),
            ('maskdcpl_mcdict_f32', Dm, Sc, Wc, {'MaxMainIter': 20, 'DataType': np.float32})):
        opt = ref_cbpdn.ConvBPDNMaskDcpl.Options(optd)
        b = ref_cbpdn.ConvBPDNMaskDcpl(DD, SS, 0.1, WW, opt)
        Y1 = b.solve()
        save(name, D=DD, S=SS, W=WW, lmbda=np.float64(0.1), Y1=Y1, X=b.X, Y=b.Y, U=b.U,
             Y0=b.var_y0(), recon=b.reconstruct(), rho_final=np.float64(b.rho),
             k_final=np.int64(b.k), **itstat_dict(b))
    for name, SS, WW, optd in (
            ('maskdcpl_f64', S, W, {'MaxMainIter': 30}),
            ('maskdcpl_f32', S, W, {'MaxMainIter': 30, 'DataType': np.float32}),
            ('maskdcpl_autorho_opts_f64', S, Wb,
             {'MaxMainIter': 30, 'rho': 2.0, 'RelaxParam': 1.5, 'NonNegCoef': True,
              'NoBndryCross': True, 'L1Weight': wl1, 'AuxVarObj': True, 'LinSolveCheck': True,
              'AutoRho': {'Enabled': True, 'Period': 3, 'Scaling': 2.0, 'RsdlRatio': 1.2,
                          'AutoScaling': True, 'RsdlTarget': 1.0}}),
            ('maskdcpl_multichan_f64', Sc, Wc, {'MaxMainIter': 20})):
        opt = ref_cbpdn.ConvBPDNMaskDcpl.Options(optd)
        b = ref_cbpdn.ConvBPDNMaskDcpl(D, SS, 0.1, WW, opt)
        Y1 = b.solve()
        extra = {'wl1': wl1} if 'L1Weight' in optd else {}
        save(name, D=D, S=SS, W=WW, lmbda=np.float64(0.1), Y1=Y1, X=b.X, Y=b.Y, U=b.U,
             recon=b.reconstruct(), rho_final=np.float64(b.rho), k_final=np.int64(b.k),
             **extra, **itstat_dict(b))


def gen_maskdl():
    """Masked dictionary learning on the ADMM mask-decoupling X-step: ConvBPDNMaskDictLearn
    (xmethod='admm', dmethod='pgm'; sporco/dictlrn/cbpdndlmd.py:219-543) and the online
    learner OnlineConvBPDNMaskDictLearn (sporco/dictlrn/onlinecdl.py:464-600)."""
    from sporco.dictlrn import cbpdndlmd as ref_md
    from sporco.dictlrn import onlinecdl as ref_online
    np.random.seed(27182)
    N, M, Nd, K = 16, 4, 5, 3
    D0 = np.random.randn(Nd, Nd, M)
    S = np.random.randn(N, N, K)
    Wd = (np.random.rand(N, N, 1, K) > 0.3).astype(np.float64)
    opt = ref_md.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 10, 'AccurateDFid': True},
                                               xmethod='admm', dmethod='pgm')
    b = ref_md.ConvBPDNMaskDictLearn(D0, S, 0.1, Wd, opt, xmethod='admm', dmethod='pgm')
    D1 = b.solve()
    save('cbpdndlmd_admm_pgm_f64', D0=D0, S=S, W=Wd, lmbda=np.float64(0.1), D1=D1,
         X=b.getcoef(), **itstat_dict(b))
    imgs = np.random.randn(N, N, 4)
    Wi = (np.random.rand(N, N, 4) > 0.3).astype(np.float64)
    Wi[..., 3] = 0.25 + 0.75 * np.random.rand(N, N)          # a non-binary weighting
    opt = ref_online.OnlineConvBPDNMaskDictLearn.Options(
        {'eta_a': 8.0, 'eta_b': 4.0, 'CBPDN': {'MaxMainIter': 30}})
    o = ref_online.OnlineConvBPDNMaskDictLearn(D0, 0.1, opt, dimK=0)
    Ds = [o.solve(imgs[..., i], Wi[..., i]).copy() for i in range(imgs.shape[-1])]
    save('onlinecdl_mask_f64', D0=D0, S=imgs, W=Wi, lmbda=np.float64(0.1), Ds=np.stack(Ds),
         **itstat_dict(o))


def gen_ccmodmd():
    """ADMM dictionary updates with mask decoupling, ConvCnstrMODMaskDcpl_IterSM / _CG
    (sporco/admm/ccmodmd.py:27-762), alone and inside ConvBPDNMaskDictLearn with the
    mask-decoupling X-step (dmethod='ism' / 'cg').  SURVEY.md 8(f) rank 3."""
    from sporco.admm import ccmodmd as ref_ccmodmd
    from sporco.dictlrn import cbpdndlmd as ref_md
    np.random.seed(16180)
    N, M, Nd, K = 16, 4, 5, 3
    S = np.random.randn(N, N, K)
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.7)
    W = (np.random.rand(N, N, 1, K) > 0.3).astype(np.float64)
    D0 = np.random.randn(Nd, Nd, M)
    autorho = {'Enabled': True, 'Period': 3, 'Scaling': 2.0, 'RsdlRatio': 1.2,
               'AutoScaling': True, 'RsdlTarget': 1.0}
    for meth, cls in (('ism', ref_ccmodmd.ConvCnstrMODMaskDcpl_IterSM),
                      ('cg', ref_ccmodmd.ConvCnstrMODMaskDcpl_CG)):
        for name, optd in (
                ('f64', {'MaxMainIter': 20}),
                ('f32', {'MaxMainIter': 20, 'DataType': np.float32}),
                ('opts_f64', {'MaxMainIter': 20, 'rho': 3.0, 'RelaxParam': 1.5,
                              'ZeroMean': True, 'LinSolveCheck': True, 'AuxVarObj': True,
                              'AutoRho': autorho})):
            if meth == 'cg':      # run CG tight: its result is then a function of its inputs
                optd = dict(optd, CG={'MaxIter': 500,
                                      'StopTol': 1e-5 if 'DataType' in optd else 1e-9})
            opt = cls.Options(optd)
            c = cls(Z, S, W, (Nd, Nd, M), opt)
            c.solve()
            save('ccmodmd_%s_%s' % (meth, name), Z=Z, S=S, W=W, dsz=np.array((Nd, Nd, M)),
                 D=c.getdict(), Y=c.Y, X=c.X, U=c.U, rho_final=np.float64(c.rho),
                 k_final=np.int64(c.k), **itstat_dict(c))
        optd = {'MaxMainIter': 10, 'AccurateDFid': True}
        if meth == 'cg':
            optd['CCMOD'] = {'CG': {'MaxIter': 500, 'StopTol': 1e-9}}
        opt = ref_md.ConvBPDNMaskDictLearn.Options(optd, xmethod='admm', dmethod=meth)
        b = ref_md.ConvBPDNMaskDictLearn(D0, S, 0.1, W, opt, xmethod='admm', dmethod=meth)
        D1 = b.solve()
        save('cbpdndlmd_admm_%s_f64' % meth, D0=D0, S=S, W=W, lmbda=np.float64(0.1), D1=D1,
             X=b.getcoef(), **itstat_dict(b))
    gen_maskdl_cg_default()
    # two-channel signal, single-channel dictionary: channels fold into the image axis in the
    # D-step and stay an axis of their own in the X-step
    Sc = np.random.randn(N, N, 2, 2)
    Wc = (np.random.rand(N, N, 2, 2) > 0.3).astype(np.float64)
    opt = ref_md.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 6, 'AccurateDFid': True},
                                               xmethod='admm', dmethod='ism')
    b = ref_md.ConvBPDNMaskDictLearn(D0, Sc, 0.1, Wc, opt, xmethod='admm', dmethod='ism')
    D1 = b.solve()
    save('cbpdndlmd_chan_admm_ism_f64', D0=D0, S=Sc, W=Wc, lmbda=np.float64(0.1), D1=D1,
         X=b.getcoef(), **itstat_dict(b))


def gen_maskdl_cg_default():
    """ConvBPDNMaskDictLearn(xmethod='admm', dmethod='cg') at the DEFAULT CG StopTol (1e-3), and
    the same reference run with linalg.inner summing the filter axis in reversed order: the rho-free
    system Z^H Z + I solved inexactly makes the outer iterates sensitive to the summation order of
    the CG operator, and the second run measures by how much (the tolerance of the test)."""
    from sporco.dictlrn import cbpdndlmd as ref_md
    rng = np.random.RandomState(1)
    N = 4
    S = rng.randn(16, 16, N)
    W = (rng.rand(16, 16, N) > 0.2).astype(np.float64)
    D0 = rng.randn(5, 5, 6)
    inner0 = ref_linalg.inner

    def inner_rev(x, y, axis=-1):
        xr = np.flip(x, axis=axis) if x.shape[axis] > 1 else x
        yr = np.flip(y, axis=axis) if y.shape[axis] > 1 else y
        return inner0(xr, yr, axis=axis)

    out = {}
    for tag, fn in (('', inner0), ('_rev', inner_rev)):
        ref_linalg.inner = fn
        try:
            opt = ref_md.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 8}, xmethod='admm', dmethod='cg')
            b = ref_md.ConvBPDNMaskDictLearn(D0, S, 0.1, W, opt, xmethod='admm', dmethod='cg')
            out['D1' + tag] = b.solve().copy()
            out['X' + tag] = b.getcoef().copy()
            if not tag:
                out.update(itstat_dict(b))
            else:
                out.update(itstat_dict(b, prefix='rev_'))
        finally:
            ref_linalg.inner = inner0
    save('cbpdndlmd_admm_cg_default_f64', D0=D0, S=S, W=W, lmbda=np.float64(0.1), **out)


def gen_ccmodmd_cns():
    """ConvCnstrMODMaskDcpl_Consensus (sporco/admm/ccmodmd.py:766-1083): the hybrid consensus /
    mask-decoupling dictionary update, alone and as dmethod='cns' of ConvBPDNMaskDictLearn with
    either X-step (sporco/dictlrn/cbpdndlmd.py:130-132, :474).  SURVEY.md 8(f) rank 3."""
    from sporco.admm import ccmodmd as ref_ccmodmd
    from sporco.dictlrn import cbpdndlmd as ref_md
    np.random.seed(14142)
    N, M, Nd, K = 16, 4, 5, 3
    S = np.random.randn(N, N, K)
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.7)
    W = (np.random.rand(N, N, 1, K) > 0.3).astype(np.float64)
    Wb = (np.random.rand(N, N) > 0.3).astype(np.float64)          # one mask for all images
    D0 = np.random.randn(Nd, Nd, M)
    autorho = {'Enabled': True, 'Period': 3, 'Scaling': 2.0, 'RsdlRatio': 1.2,
               'AutoScaling': True, 'RsdlTarget': 1.0}
    cls = ref_ccmodmd.ConvCnstrMODMaskDcpl_Consensus
    for name, WW, optd in (
            ('f64', W, {'MaxMainIter': 20}),
            ('f32', W, {'MaxMainIter': 20, 'DataType': np.float32}),
            ('opts_f64', Wb, {'MaxMainIter': 20, 'rho': 3.0, 'RelaxParam': 1.5,
                              'ZeroMean': True, 'AutoRho': autorho}),
            ('std_f64', W, {'MaxMainIter': 20, 'AbsStopTol': 1e-4, 'RelStopTol': 1e-3,
                            'AutoRho': dict(autorho, StdResiduals=True, AutoScaling=False)})):
        opt = cls.Options(optd)
        c = cls(Z, S, WW, (Nd, Nd, M), opt)
        c.solve()
        save('ccmodmd_cns_%s' % name, Z=Z, S=S, W=WW, dsz=np.array((Nd, Nd, M)),
             D=c.getdict(), Y=c.Y, X=c.X, U=c.U, Y1=c.Y1, U1=c.U1, rho_final=np.float64(c.rho),
             k_final=np.int64(c.k), **itstat_dict(c))
    for xm in ('admm', 'pgm'):
        opt = ref_md.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 10, 'AccurateDFid': True},
                                                   xmethod=xm, dmethod='cns')
        b = ref_md.ConvBPDNMaskDictLearn(D0, S, 0.1, W, opt, xmethod=xm, dmethod='cns')
        D1 = b.solve()
        save('cbpdndlmd_%s_cns_f64' % xm, D0=D0, S=S, W=W, lmbda=np.float64(0.1), D1=D1,
             X=b.getcoef(), **itstat_dict(b))
    # two-channel signal, single-channel dictionary (channels fold into the consensus blocks)
    Sc = np.random.randn(N, N, 2, 2)
    Wc = (np.random.rand(N, N, 2, 2) > 0.3).astype(np.float64)
    Zc = np.random.randn(N, N, 2, 2, M) * (np.random.rand(N, N, 2, 2, M) > 0.7)
    c = cls(Zc, Sc, Wc, (Nd, Nd, M), cls.Options({'MaxMainIter': 12}))
    c.solve()
    save('ccmodmd_cns_chan_f64', Z=Zc, S=Sc, W=Wc, dsz=np.array((Nd, Nd, M)), D=c.getdict(),
         Y=c.Y, rho_final=np.float64(c.rho), k_final=np.int64(c.k), **itstat_dict(c))


def gen_signal():
    """Pre/post-processing around the solver: signal.tikhonov_filter
    (sporco/signal.py:244-301), fft.fftconv (sporco/fft.py:376-417), signal.gradient_filters."""
    from sporco import signal as ref_signal
    np.random.seed(24680)
    s2 = np.random.randn(20, 17)
    s3 = np.random.randn(16, 16, 3).astype(np.float32)
    sl2, sh2 = ref_signal.tikhonov_filter(s2, 5.0, 16)
    sl3, sh3 = ref_signal.tikhonov_filter(s3, 2.0, 4)
    d = np.random.randn(5, 5, 4)
    x = np.random.randn(16, 12, 4)
    cv = ref_fft.fftconv(d, x, axes=(0, 1), origin=(2, 2))
    cv1 = ref_fft.fftconv(np.random.RandomState(1).randn(3, 3), s2)
    Gf, GHGf = ref_signal.gradient_filters(5, (0, 1), (12, 9), dtype=np.dtype(np.float64))
    save('signal_prims', s2=s2, s3=s3, sl2=sl2, sh2=sh2, sl3=sl3, sh3=sh3, d=d, x=x, cv=cv,
         k3=np.random.RandomState(1).randn(3, 3), cv1=cv1, Gf=Gf, GHGf=GHGf)


def gen_mask():
    """Masked data fidelity by PGM: pgm.cbpdn.ConvBPDNMask (sporco/pgm/cbpdn.py:387-506) and
    pgm.ccmod.ConvCnstrMODMask (sporco/pgm/ccmod.py:408-604).  SURVEY.md 8(f) rank 3."""
    np.random.seed(86420)
    N, M, Nd, K = 16, 4, 5, 3
    D = np.random.randn(Nd, Nd, M)
    S = np.random.randn(N, N, K)
    W = (np.random.rand(N, N, K) > 0.3).astype(np.float64)
    W1 = (np.random.rand(N, N) > 0.3).astype(np.float64)
    for name, Wm, optd in (('pgm_mask_f64', W, {'MaxMainIter': 30, 'L': 500.0}),
                           ('pgm_mask_f32', W, {'MaxMainIter': 30, 'L': 500.0,
                                                'DataType': np.float32}),
                           ('pgm_mask_bcast_bt_f64', W1.reshape(N, N, 1),
                            {'MaxMainIter': 25, 'L': 1.0, 'Backtrack': BacktrackStandard()})):
        optd = dict(optd, RelStopTol=0.0)
        opt = ref_pgm_cbpdn.ConvBPDNMask.Options(optd)
        b = ref_pgm_cbpdn.ConvBPDNMask(D, S, 0.1, Wm, opt)
        b.solve()
        save(name, D=D, S=S, W=Wm, lmbda=np.float64(0.1), X=b.X, Xf=b.Xf,
             L_final=np.float64(b.L), k_final=np.int64(b.k), **itstat_dict(b))
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.7)
    opt = ref_pgm_ccmod.ConvCnstrMODMask.Options({'MaxMainIter': 20, 'L': 800.0})
    Wi = W.reshape(N, N, 1, K, 1)      # internal layout of S (pgm/ccmod.py:480-486)
    c = ref_pgm_ccmod.ConvCnstrMODMask(Z, S, Wi, (Nd, Nd, M), opt)
    c.solve()
    from sporco.dictlrn import cbpdndlmd as ref_md
    D0 = np.random.randn(Nd, Nd, M)
    Wd = W.reshape(N, N, 1, K)
    opt = ref_md.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 10, 'AccurateDFid': True},
                                               xmethod='pgm', dmethod='pgm')
    b = ref_md.ConvBPDNMaskDictLearn(D0, S, 0.1, Wd, opt, xmethod='pgm', dmethod='pgm')
    D1 = b.solve()
    save('cbpdndlmd_pgm_f64', D0=D0, S=S, W=Wd, lmbda=np.float64(0.1), D1=D1, X=b.getcoef(),
         **itstat_dict(b))
    save('pgm_ccmod_mask_f64', Z=Z, S=S, W=Wi, dsz=np.array((Nd, Nd, M)), D=c.getdict(),
         Xfull=c.X, **itstat_dict(c))


def gen_mask_mcdict():
    """The masked PGM classes with a multi-channel dictionary (pgm.cbpdn.ConvBPDNMask
    sporco/pgm/cbpdn.py:387-506, pgm.ccmod.ConvCnstrMODMask sporco/pgm/ccmod.py:408-631; the
    reference's dictlrn tests run both with colour dictionaries)."""
    np.random.seed(112233)
    N, C, K, M, Nd = 16, 3, 2, 4, 5
    D = np.random.randn(Nd, Nd, C, M)
    S = np.random.randn(N, N, C, K)
    W = (np.random.rand(N, N, C, K) > 0.3).astype(np.float64)
    Wb = (np.random.rand(N, N, 1, K) > 0.3).astype(np.float64)
    for name, w, optd in (('pgm_mask_mcdict_f64', W, {'MaxMainIter': 15, 'L': 100.0}),
                          ('pgm_mask_mcdict_bcast_f32', Wb, {'MaxMainIter': 15, 'L': 100.0,
                                                             'DataType': np.float32})):
        b = ref_pgm_cbpdn.ConvBPDNMask(D, S, 0.1, w, ref_pgm_cbpdn.ConvBPDNMask.Options(optd))
        b.solve()
        save(name, D=D, S=S, W=w, lmbda=np.float64(0.1), X=b.getcoef(), recon=b.reconstruct(),
             **itstat_dict(b))
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.6)
    c = ref_pgm_ccmod.ConvCnstrMODMask(Z, S, W, (Nd, Nd, C, M),
                                       ref_pgm_ccmod.ConvCnstrMODMask.Options({'MaxMainIter': 15, 'L': 50.0}))
    c.solve()
    save('pgm_ccmod_mask_mcdict_f64', Z=Z, S=S, W=W, dsz=np.array((Nd, Nd, C, M)), D=c.getdict(),
         **itstat_dict(c))


def gen_multiscale():
    """Multi-scale dictionaries (a `dsz` of size blocks: cnvrep.bcrop / zeromean / Pcn,
    sporco/cnvrep.py:609-670, :729-817, :868-913) in the dictionary updates and in dictionary
    learning -- the reference's examples/scripts/cdl/cbpdndl_pgm_clr.py learns a multi-scale
    colour dictionary through the DictSize option."""
    np.random.seed(424242)
    N, M, K, C = 16, 8, 2, 3
    dsz = ((4, 4, 3), (6, 6, 2), (8, 8, 3))
    v = np.random.randn(N, N, 1, 1, M)
    save('cnvrep_multiscale', v=v, dsz=np.array(dsz), bcrop=ref_cnvrep.bcrop(v, dsz),
         zeromean=ref_cnvrep.zeromean(v, dsz),
         pcn=ref_cnvrep.Pcn(v, dsz, (N, N), 2, 1, crp=False, zm=True),
         pcn_crop=ref_cnvrep.Pcn(v, dsz, (N, N), 2, 1, crp=True, zm=False))
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.6)
    S = np.random.randn(N, N, K)
    c = ref_pgm_ccmod.ConvCnstrMOD(Z, S, dsz, ref_pgm_ccmod.ConvCnstrMOD.Options(
        {'MaxMainIter': 15, 'ZeroMean': True, 'L': 50.0}))
    c.solve()
    save('pgm_ccmod_multiscale_f64', Z=Z, S=S, dsz=np.array(dsz), D=c.getdict(), **itstat_dict(c))
    c = ref_admm_ccmod.ConvCnstrMOD_Consensus(Z, S, dsz, ref_admm_ccmod.ConvCnstrMOD_Consensus.Options(
        {'MaxMainIter': 12}))
    c.solve()
    save('ccmod_cns_multiscale_f64', Z=Z, S=S, dsz=np.array(dsz), D=c.getdict(), Y=c.Y,
         **itstat_dict(c))
    dszc = ((4, 4, C, 3), (6, 6, C, 2), (8, 8, C, 3))
    D0 = np.random.randn(8, 8, C, M)
    Sc = np.random.randn(N, N, C, K)
    opt = ref_cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 8, 'DictSize': dszc, 'CBPDN': {'L': 50.0}, 'CCMOD': {'L': 50.0}},
        xmethod='pgm', dmethod='pgm')
    b = ref_cbpdndl.ConvBPDNDictLearn(D0, Sc, 0.1, opt, xmethod='pgm', dmethod='pgm')
    D1 = b.solve()
    save('cbpdndl_multiscale_clr_f64', D0=D0, S=Sc, dsz=np.array(dszc), lmbda=np.float64(0.1), D1=D1,
         X=b.getcoef(), **itstat_dict(b))


def gen_ams():
    """AddMaskSim (sporco/admm/cbpdn.py:2287-2485) around ConvBPDN, ConvBPDNJoint and
    ConvBPDNGradReg: SURVEY.md 8(f) rank 1."""
    np.random.seed(97531)
    D = np.random.randn(5, 5, 3)
    S = np.random.randn(16, 16, 2)
    W2 = (np.random.rand(16, 16) > 0.3).astype(np.float64)      # broadcast over images
    W3 = (np.random.rand(16, 16, 2) > 0.3).astype(np.float64)   # one mask per image
    ams_case('ams_cbpdn_f64', ref_cbpdn.ConvBPDN, D, S, W3, (0.1,), {'MaxMainIter': 25})
    ams_case('ams_cbpdn_f32', ref_cbpdn.ConvBPDN, D, S, W3, (0.1,),
             {'MaxMainIter': 25, 'DataType': np.float32})
    ams_case('ams_cbpdn_bcast_nonneg_f64', ref_cbpdn.ConvBPDN, D, S, W2, (0.1,),
             {'MaxMainIter': 20, 'NonNegCoef': True, 'NoBndryCross': True,
              'AuxVarObj': True})
    ams_case('ams_gradreg_f64', ref_cbpdn.ConvBPDNGradReg, D, S, W3, (0.1, 0.2),
             {'MaxMainIter': 20, 'GradWeight': np.array([0.0, 1.0, 0.5, 1.0])})
    Sc = np.random.randn(16, 16, 3, 2)
    Wc = (np.random.rand(16, 16, 3, 2) > 0.3).astype(np.float64)
    ams_case('ams_joint_f64', ref_cbpdn.ConvBPDNJoint, D, Sc, Wc, (0.1, 0.05),
             {'MaxMainIter': 20})

def gen_zchan():
    """Dictionary updates of a multi-channel dictionary whose coefficient maps carry the
    channels too -- (N, N, Nc, K, M) maps with an (Nd, Nd, Nc, M) dictionary -- which the
    reference's broadcasting turns into Nc single-channel updates sharing rho, the step size
    and the residuals (its own tests: tests/admm/test_ccmod.py:278-295, tests/pgm/test_ccmod.py
    :175-191)."""
    np.random.seed(24680)
    N, M, K, Nc, Nd = 16, 4, 2, 3, 8
    Z = np.random.randn(N, N, Nc, K, M) * (np.random.rand(N, N, Nc, K, M) > 0.5)
    S = np.random.randn(N, N, Nc, K)
    cls = ref_admm_ccmod.ConvCnstrMOD_Consensus
    for name, optd in (('ccmod_cns_zchan_f64', {'MaxMainIter': 20, 'LinSolveCheck': True}),
                       ('ccmod_cns_zchan_opts_f32', {'MaxMainIter': 12, 'ZeroMean': True,
                                                     'AuxVarObj': False, 'DataType': np.float32})):
        c = cls(Z, S, (Nd, Nd, Nc, M), cls.Options(optd))
        c.solve()
        save(name, Z=Z, S=S, dsz=np.array((Nd, Nd, Nc, M)), D=c.getdict(), Y=c.Y, U=c.U, X=c.X,
             rho_final=np.float64(c.rho), k_final=np.int64(c.k),
             **itstat_dict(c))
    c = ref_pgm_ccmod.ConvCnstrMOD(Z, S, (Nd, Nd, Nc, M),
                                   ref_pgm_ccmod.ConvCnstrMOD.Options({'MaxMainIter': 20, 'L': 400.0}))
    c.solve()
    save('pgm_ccmod_zchan_f64', Z=Z, S=S, dsz=np.array((Nd, Nd, Nc, M)), D=c.getdict(), X=c.X,
         **itstat_dict(c))
    # the masked PGM update with such maps (tests/pgm/test_ccmod.py:546-563), and a dsz with an
    # explicit single channel, which makes the third axis of S channels, not images (:453-468)
    W = np.random.randn(N, N, Nc, K)
    cls = ref_pgm_ccmod.ConvCnstrMODMask
    c = cls(Z, S, W, (Nd, Nd, Nc, M), cls.Options({'MaxMainIter': 20, 'L': 400.0}))
    c.solve()
    save('pgm_ccmod_mask_zchan_f64', Z=Z, S=S, W=W, dsz=np.array((Nd, Nd, Nc, M)), D=c.getdict(),
         X=c.X, **itstat_dict(c))
    Z1 = np.random.randn(N, N, 1, 3, M)
    S1 = np.random.randn(N, N, 3)
    W1 = np.random.randn(N, N, 3)
    c = cls(Z1, S1, W1, (Nd, Nd, 1, M), cls.Options({'MaxMainIter': 20, 'L': 400.0}))
    c.solve()
    save('pgm_ccmod_mask_dsz1chan_f64', Z=Z1, S=S1, W=W1, dsz=np.array((Nd, Nd, 1, M)),
         D=c.getdict(), X=c.X, **itstat_dict(c))

def gen_ccmodmd_cns_mcdict():
    """ConvCnstrMODMaskDcpl_Consensus with a multi-channel (colour) dictionary
    (sporco/admm/ccmodmd.py:766-1083 on ccmod.py:696-698): channel-less coefficient maps as the
    sparse coding step produces them, maps that carry the channels (the reference's own
    tests/admm/test_ccmodmd.py:333-351), and masked colour dictionary learning with
    dmethod='cns' (the reference's examples/scripts/cdl/cbpdndl_md_clr.py in miniature)."""
    from sporco.admm import ccmodmd as ref_ccmodmd
    from sporco.dictlrn import cbpdndlmd as ref_md
    np.random.seed(57721)
    N, M, Nd, K, Nc = 16, 4, 5, 3, 3
    S = np.random.randn(N, N, Nc, K)
    Z = np.random.randn(N, N, 1, K, M) * (np.random.rand(N, N, 1, K, M) > 0.6)
    Zc = np.random.randn(N, N, Nc, K, M) * (np.random.rand(N, N, Nc, K, M) > 0.6)
    W = (np.random.rand(N, N, Nc, K) > 0.3).astype(np.float64)
    Wb = (np.random.rand(N, N) > 0.3).astype(np.float64)
    cls = ref_ccmodmd.ConvCnstrMODMaskDcpl_Consensus
    for name, ZZ, WW, optd in (
            ('f64', Z, W, {'MaxMainIter': 15, 'LinSolveCheck': True}),
            ('opts_f32', Z, Wb, {'MaxMainIter': 15, 'rho': 3.0, 'RelaxParam': 1.5,
                                 'ZeroMean': True, 'DataType': np.float32}),
            ('zchan_f64', Zc, Wb, {'MaxMainIter': 15, 'LinSolveCheck': True})):
        c = cls(ZZ, S, WW, (Nd, Nd, Nc, M), cls.Options(optd))
        c.solve()
        save('ccmodmd_cns_mcdict_%s' % name, Z=ZZ, S=S, W=WW, dsz=np.array((Nd, Nd, Nc, M)),
             D=c.getdict(), Y=c.Y, X=c.X, U=c.U, Y1=c.Y1, U1=c.U1, rho_final=np.float64(c.rho),
             k_final=np.int64(c.k), **itstat_dict(c))
    D0 = np.random.randn(Nd, Nd, Nc, M)
    Wd = (np.random.rand(N, N, 1, K) > 0.3).astype(np.float64)
    opt = ref_md.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                               xmethod='admm', dmethod='cns')
    b = ref_md.ConvBPDNMaskDictLearn(D0, S, 0.1, Wd, opt, xmethod='admm', dmethod='cns')
    D1 = b.solve()
    save('cbpdndlmd_admm_cns_mcdict_f64', D0=D0, S=S, W=Wd, lmbda=np.float64(0.1), D1=D1,
         X=b.getcoef(), **itstat_dict(b))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['primitives', 'admm', 'known', 'config1', 'pgm',
                             'pcn', 'dictlearn', 'gradreg', 'ams', 'mcdict', 'mcdict_classes', 'cns', 'cns_options', 'cns_mcdict', 'ccmod_eq', 'ccmod_ism_many', 'online', 'shard', 'maskdcpl', 'maskdl', 'ccmodmd', 'ccmodmd_cns', 'shard_cns', 'signal', 'mask', 'mask_mcdict', 'multiscale', 'zchan', 'ccmodmd_cns_mcdict', 'ccmod_eq_mcdict']
    table = {'primitives': gen_primitives, 'admm': gen_admm, 'gradreg': gen_gradreg, 'ams': gen_ams, 'mcdict': gen_mcdict, 'mcdict_classes': gen_mcdict_classes, 'cns': gen_cns, 'cns_options': gen_cns_options, 'cns_mcdict': gen_cns_mcdict, 'ccmod_eq': gen_ccmod_eq, 'ccmod_ism_many': gen_ccmod_ism_many, 'online': gen_online, 'shard': gen_shard, 'maskdcpl': gen_maskdcpl, 'maskdl': gen_maskdl, 'ccmodmd': gen_ccmodmd, 'signal': gen_signal, 'mask': gen_mask, 'mask_mcdict': gen_mask_mcdict, 'multiscale': gen_multiscale, 'zchan': gen_zchan, 'ccmodmd_cns_mcdict': gen_ccmodmd_cns_mcdict,
             'known': gen_known_answer, 'config1': gen_config1, 'dim1': gen_dim1, 'dim1_dl': gen_dim1_dl, 'dim3': gen_dim3, 'dim3_dl': gen_dim3_dl, 'dim1_dstep': gen_dim1_dstep, 'ccmod_cplx': gen_ccmod_cplx,
             'config2': gen_config2, 'mixed_radix': gen_mixed_radix, 'tol': gen_tol, 'config5': gen_config5,
             'config3': gen_config3, 'config4': gen_config4, 'ccmod_eq_mcdict': gen_ccmod_eq_mcdict,
             'ccmodmd_cns': gen_ccmodmd_cns, 'shard_cns': gen_shard_cns,
             'admm_cplx': gen_admm_cplx, 'pgm': gen_pgm, 'pgm_bt256': gen_pgm_bt256, 'pgm_btrobust256': gen_pgm_btrobust256, 'pgm_monotone256': gen_pgm_monotone256, 'pgm_stepsize256': gen_pgm_stepsize256, 'maskdl_cg_default': gen_maskdl_cg_default, 'pcn': gen_pcn, 'dictlearn': gen_dictlearn}
    for w in which:
        table[w]()
