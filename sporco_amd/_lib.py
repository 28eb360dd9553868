"""ctypes binding of libsporco_amd.so (the C ABI declared in include/sporco_amd.h).

There is exactly one compute path: the HIP library built for gfx950.  If the
shared object is missing, or no AMD GPU is visible, the functions here raise
:class:`BackendError`; nothing falls back to NumPy.
"""

import ctypes
import os

import numpy as np

F32, F64 = 0, 1

VAR_Y, VAR_U, VAR_X, VAR_XF, VAR_DF, VAR_SF = 0, 1, 2, 3, 4, 5
VAR_YF, VAR_XFPRV, VAR_YFPRV, VAR_VF, VAR_GF, VAR_AX, VAR_YPREV = 6, 7, 8, 9, 10, 11, 12
VAR_T0, VAR_T1, VAR_T2, VAR_ZF = 13, 14, 15, 16
VAR_CX, VAR_CU = 17, 18
VAR_MY0, VAR_MU0 = 19, 20
VAR_DMY0, VAR_DMU0 = 21, 22
VAR_DX, VAR_DXF, VAR_DYF, VAR_DXFPRV, VAR_DYFPRV = 32, 33, 34, 35, 36
VAR_DVF, VAR_DGF, VAR_DT0, VAR_DT1, VAR_DT2 = 37, 38, 39, 40, 41
VAR_DSX, VAR_DSU = 42, 43
DSTEP_ISM, DSTEP_CG = 0, 1

FLAG_NONNEG = 1 << 0
FLAG_NOBNDRY = 1 << 1
FLAG_JOINT = 1 << 2
FLAG_RESID = 1 << 3
FLAG_OBJ = 1 << 4
FLAG_XRRS = 1 << 5
FLAG_GEVAL_Y = 1 << 6
FLAG_FEVAL_Y = 1 << 7
FLAG_KEEP_X = 1 << 8
FLAG_NO_X = 1 << 9
FLAG_GRADREG = 1 << 10
FLAG_AMS = 1 << 11
FLAG_DMASK = 1 << 12
QUERY_FUSED_COLS, QUERY_FUSED_ROWS, QUERY_FUSED_PGM, QUERY_DEVICE_FILTERS, QUERY_VFORM_LIVE = 0, 1, 2, 3, 4
QUERY_PERSIST_RUNS = 5
QUERY_CCMOD_GROUPS = 6
HINT_KEEP_VFORM = 0
HINT_ONE_LAUNCH = 1
MODE_COMPLEX_PAIR = 2
VOLUME_FILTER_DEPTH = 3

OUT_R2, OUT_S2, OUT_AX2, OUT_Y2, OUT_U2 = 0, 1, 2, 3, 4
OUT_DFID, OUT_L1, OUT_L21 = 5, 6, 7
OUT_XRRS_D2, OUT_XRRS_AX2, OUT_XRRS_B2 = 8, 9, 10
OUT_RGR = 11
OUT_CNSTR = 12
OUT_CGIT, OUT_CGN = 13, 14
OUT_COUNT = 16

PGM_F, PGM_DFID, PGM_L1, PGM_HESS, PGM_RSDL, PGM_FY, PGM_LIN, PGM_DXY2 = range(8)

EXPORTS = (
    'sporco_amd_version', 'sporco_amd_last_error', 'sporco_amd_device_count',
    'sporco_amd_device_info', 'sporco_amd_csc_create', 'sporco_amd_csc_create_mc', 'sporco_amd_csc_create_volume',
    'sporco_amd_csc_destroy',
    'sporco_amd_csc_sync', 'sporco_amd_csc_stream', 'sporco_amd_csc_query', 'sporco_amd_csc_placement_report', 'sporco_amd_csc_set_hint', 'sporco_amd_csc_set_signal', 'sporco_amd_csc_set_dict', 'sporco_amd_csc_set_dict_imag',
    'sporco_amd_csc_set_l1_weight', 'sporco_amd_csc_set_l21_weight',
    'sporco_amd_csc_set_grad_weight', 'sporco_amd_csc_set_ams_mask',
    'sporco_amd_csc_upload', 'sporco_amd_csc_download', 'sporco_amd_csc_device_ptr',
    'sporco_amd_csc_admm_iter', 'sporco_amd_csc_admm_iter_dev', 'sporco_amd_csc_admm_run',
    'sporco_amd_csc_admm_xstep', 'sporco_amd_csc_admm_relax',
    'sporco_amd_csc_admm_ystep', 'sporco_amd_csc_admm_ustep',
    'sporco_amd_csc_admm_stats', 'sporco_amd_csc_scale_u',
    'sporco_amd_csc_reconstruct', 'sporco_amd_csc_dhs_absmax',
    'sporco_amd_csc_pgm_grad', 'sporco_amd_csc_pgm_eval', 'sporco_amd_csc_pgm_prox_step',
    'sporco_amd_csc_pgm_iter', 'sporco_amd_csc_pgm_commit',
    'sporco_amd_csc_lincomb', 'sporco_amd_csc_pair_stats', 'sporco_amd_csc_copy',
    'sporco_amd_csc_pgm_resid', 'sporco_amd_csc_pgm_resid_stats',
    'sporco_amd_csc_fft_var', 'sporco_amd_csc_ifft_var',
    'sporco_amd_csc_ccmod_setcoef', 'sporco_amd_csc_ccmod_grad', 'sporco_amd_csc_ccmod_eval',
    'sporco_amd_csc_ccmod_prox_step', 'sporco_amd_csc_ccmod_cnstr',
    'sporco_amd_csc_ccmod_getdict', 'sporco_amd_csc_setdict_from_dstep', 'sporco_amd_csc_asum',
    'sporco_amd_csc_set_filter_sizes', 'sporco_amd_csc_cns_init', 'sporco_amd_csc_cns_iter', 'sporco_amd_csc_cns_md_init', 'sporco_amd_csc_cns_mean_ptr',
    'sporco_amd_csc_dstep_init', 'sporco_amd_csc_dstep_iter', 'sporco_amd_csc_ccmod_sgd_step',
    'sporco_amd_csc_mdcpl_init', 'sporco_amd_csc_mdcpl_iter', 'sporco_amd_csc_dstep_md_init',
    'sporco_amd_csc_set_data_mask', 'sporco_amd_csc_masked_grad',
    'sporco_amd_csc_profile', 'sporco_amd_csc_profile_read', 'sporco_amd_profile_slots',
    'sporco_amd_dev_malloc', 'sporco_amd_dev_free', 'sporco_amd_dev_upload',
    'sporco_amd_dev_download', 'sporco_amd_dev_axpby', 'sporco_amd_tikhonov_filter_dev',
    'sporco_amd_fftconv_dev', 'sporco_amd_csc_set_signal_dev', 'sporco_amd_csc_reconstruct_dev',
    'sporco_amd_transfer_stats',
    'sporco_amd_comm_unique_id', 'sporco_amd_comm_create', 'sporco_amd_comm_destroy',
    'sporco_amd_comm_info', 'sporco_amd_comm_allreduce', 'sporco_amd_comm_allreduce_host',
    'sporco_amd_csc_set_comm',
    'sporco_amd_rfftn2', 'sporco_amd_irfftn2', 'sporco_amd_solvedbi_sm',
    'sporco_amd_inner', 'sporco_amd_prox_l1', 'sporco_amd_prox_l1w', 'sporco_amd_prox_sl1l2',
    'sporco_amd_rfl2norm2',
)


class BackendError(RuntimeError):
    """The HIP backend is missing, failed to load, or reported an error."""


class Dims(ctypes.Structure):
    _fields_ = [('H', ctypes.c_int32), ('W', ctypes.c_int32), ('C', ctypes.c_int32),
                ('N', ctypes.c_int32), ('K', ctypes.c_int32), ('dtype', ctypes.c_int32)]


class PgmParams(ctypes.Structure):
    _fields_ = [('L', ctypes.c_double), ('lmbda', ctypes.c_double), ('beta', ctypes.c_double),
                ('flags', ctypes.c_uint32), ('dH', ctypes.c_int32), ('dW', ctypes.c_int32),
                ('want_stats', ctypes.c_int32), ('hold', ctypes.c_int32)]


class CnsParams(ctypes.Structure):
    _fields_ = [('rho', ctypes.c_double), ('rlx', ctypes.c_double), ('u_scale', ctypes.c_double),
                ('flags', ctypes.c_uint32), ('dH', ctypes.c_int32), ('dW', ctypes.c_int32),
                ('zero_mean', ctypes.c_int32), ('mask_dcpl', ctypes.c_int32),
                ('phase', ctypes.c_int32)]


class DstepParams(ctypes.Structure):
    _fields_ = [('rho', ctypes.c_double), ('rlx', ctypes.c_double), ('u_scale', ctypes.c_double),
                ('cg_tol', ctypes.c_double), ('flags', ctypes.c_uint32), ('dH', ctypes.c_int32),
                ('dW', ctypes.c_int32), ('zero_mean', ctypes.c_int32), ('method', ctypes.c_int32),
                ('cg_maxiter', ctypes.c_int32), ('mask_dcpl', ctypes.c_int32)]


class AdmmCtrl(ctypes.Structure):
    """sporco_amd_admm_ctrl (include/sporco_amd.h): constants of a device-driven solve."""
    _fields_ = [('abs_tol', ctypes.c_double), ('rel_tol', ctypes.c_double),
                ('sqrt_nc', ctypes.c_double), ('sqrt_nx', ctypes.c_double),
                ('rho_tau', ctypes.c_double), ('rho_mu', ctypes.c_double),
                ('rho_xi', ctypes.c_double),
                ('auto_rho', ctypes.c_int32), ('period', ctypes.c_int32),
                ('auto_scaling', ctypes.c_int32), ('std_residuals', ctypes.c_int32),
                ('need_residuals', ctypes.c_int32), ('k0', ctypes.c_int32),
                ('max_iter', ctypes.c_int32), ('lookahead', ctypes.c_int32)]


class AdmmRecord(ctypes.Structure):
    """sporco_amd_admm_record: what one iteration of a device-driven solve leaves behind."""
    _fields_ = [('sums', ctypes.c_double * 16), ('r', ctypes.c_double), ('s', ctypes.c_double),
                ('epri', ctypes.c_double), ('edua', ctypes.c_double), ('rho', ctypes.c_double),
                ('u_scale', ctypes.c_double), ('seconds', ctypes.c_double),
                ('k', ctypes.c_int32), ('stop', ctypes.c_int32)]


REDUCE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)
EUNSUPPORTED = -5


class AdmmParams(ctypes.Structure):
    _fields_ = [('rho', ctypes.c_double), ('lmbda', ctypes.c_double),
                ('mu', ctypes.c_double), ('rlx', ctypes.c_double),
                ('u_scale', ctypes.c_double), ('flags', ctypes.c_uint32),
                ('dH', ctypes.c_int32), ('dW', ctypes.c_int32)]


_DEFAULT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                             'libsporco_amd.so')
_lib = None
_lib_path = None


def default_library_path():
    return _DEFAULT_PATH


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) and
    link it by file name, so a process that loads this library (which needs the SONAME
    from /opt/rocm) *and* torch ends up with two HIP runtimes, and the second one to
    initialise finds no GPU.  torch.distributed is how the multi-GPU path talks to
    RCCL, so when torch is installed its copy is loaded first: the dynamic linker then
    resolves our dependency to it by SONAME and a later ``import torch`` reuses the
    same object.  SPORCO_AMD_SHARE_TORCH_RUNTIME=0 disables this."""
    if os.environ.get('SPORCO_AMD_SHARE_TORCH_RUNTIME', '1') == '0':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass     # fall back to the system runtime


def load(path=None):
    """Load the backend shared library (default: the in-tree hipcc build).

    ``path`` exists so that the test-suite can point the binding at the CPU
    fiber simulator build of the same sources (tests/hostsim); product code
    never passes it.
    """
    global _lib, _lib_path
    # (SPORCO_AMD_LIBRARY: an alternative hipcc build of the same sources, for A/B timing)
    path = os.path.abspath(path or os.environ.get('SPORCO_AMD_LIBRARY') or _DEFAULT_PATH)
    if not os.path.exists(path):
        raise BackendError(
            "sporco_amd: HIP library %s not found. Build it with "
            "`make -C sporco_amd/csrc` (or `python -c 'import __graft_entry__ as g; "
            "g.build()'`). There is no CPU fallback." % path)
    _share_torch_hip_runtime()
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise BackendError("sporco_amd: cannot load %s: %s" % (path, e))
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise BackendError("sporco_amd: %s lacks symbols %s" % (path, missing))
    lib.sporco_amd_version.restype = ctypes.c_char_p
    lib.sporco_amd_last_error.restype = ctypes.c_char_p
    for name in EXPORTS:
        if name not in ('sporco_amd_version', 'sporco_amd_last_error'):
            getattr(lib, name).restype = ctypes.c_int
    lib.sporco_amd_csc_create.argtypes = [ctypes.POINTER(Dims), ctypes.c_int,
                                          ctypes.c_void_p,
                                          ctypes.POINTER(ctypes.c_void_p)]
    lib.sporco_amd_csc_create_mc.argtypes = [ctypes.POINTER(Dims), ctypes.c_int32, ctypes.c_int,
                                             ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_void_p)]
    lib.sporco_amd_csc_create_volume.argtypes = lib.sporco_amd_csc_create_mc.argtypes
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
    dptr = ctypes.POINTER(ctypes.c_double)
    pptr = ctypes.POINTER(AdmmParams)
    sig = {
        'sporco_amd_device_count': [ctypes.POINTER(ctypes.c_int)],
        'sporco_amd_device_info': [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_size_t)],
        'sporco_amd_csc_destroy': [vp], 'sporco_amd_csc_sync': [vp],
        'sporco_amd_csc_stream': [vp, ctypes.POINTER(ctypes.c_void_p)],
        'sporco_amd_csc_query': [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)],
        'sporco_amd_csc_placement_report': [vp, ctypes.c_char_p, ctypes.c_size_t],
        'sporco_amd_csc_set_hint': [vp, ctypes.c_int, ctypes.c_int],
        'sporco_amd_csc_set_signal': [vp, vp],
        'sporco_amd_csc_set_dict': [vp, vp, i32, i32],
        'sporco_amd_csc_set_dict_imag': [vp, vp, i32, i32],
        'sporco_amd_csc_set_l1_weight': [vp, vp, ctypes.POINTER(i64)],
        'sporco_amd_csc_set_l21_weight': [vp, vp, ctypes.POINTER(i64)],
        'sporco_amd_csc_set_grad_weight': [vp, vp],
        'sporco_amd_csc_set_filter_sizes': [vp, ctypes.POINTER(ctypes.c_int32),
                                            ctypes.POINTER(ctypes.c_int32)],
        'sporco_amd_csc_set_ams_mask': [vp, vp, ctypes.POINTER(i64)],
        'sporco_amd_csc_upload': [vp, ctypes.c_int, vp],
        'sporco_amd_csc_download': [vp, ctypes.c_int, vp],
        'sporco_amd_csc_device_ptr': [vp, ctypes.c_int, ctypes.POINTER(vp)],
        'sporco_amd_csc_admm_iter': [vp, pptr, dptr],
        'sporco_amd_csc_admm_iter_dev': [vp, pptr, vp],
        'sporco_amd_csc_admm_run': [vp, pptr, ctypes.POINTER(AdmmCtrl), ctypes.POINTER(AdmmRecord),
                                    ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(dbl),
                                    ctypes.POINTER(dbl), REDUCE_FN, vp],
        'sporco_amd_csc_admm_xstep': [vp, pptr, dptr],
        'sporco_amd_csc_admm_relax': [vp, dbl],
        'sporco_amd_csc_admm_ystep': [vp, pptr],
        'sporco_amd_csc_admm_ustep': [vp, pptr],
        'sporco_amd_csc_admm_stats': [vp, pptr, dptr],
        'sporco_amd_csc_scale_u': [vp, dbl],
        'sporco_amd_csc_reconstruct': [vp, ctypes.c_int, vp],
        'sporco_amd_csc_dhs_absmax': [vp, dptr],
        'sporco_amd_csc_pgm_grad': [vp, ctypes.c_int, dptr],
        'sporco_amd_csc_pgm_eval': [vp, ctypes.c_int, dptr],
        'sporco_amd_csc_pgm_iter': [vp, ctypes.POINTER(PgmParams), dptr],
        'sporco_amd_csc_pgm_commit': [vp],
        'sporco_amd_csc_pgm_prox_step': [vp, dbl, dbl, ctypes.c_uint32, i32, i32, dptr],
        'sporco_amd_csc_lincomb': [vp, ctypes.c_int, dbl, ctypes.c_int, dbl, ctypes.c_int, dbl,
                                   ctypes.c_int],
        'sporco_amd_csc_pair_stats': [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dptr],
        'sporco_amd_csc_pgm_resid': [vp, ctypes.c_int, ctypes.c_int],
        'sporco_amd_csc_pgm_resid_stats': [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dptr],
        'sporco_amd_csc_copy': [vp, ctypes.c_int, ctypes.c_int],
        'sporco_amd_csc_fft_var': [vp, ctypes.c_int, ctypes.c_int],
        'sporco_amd_csc_ifft_var': [vp, ctypes.c_int, ctypes.c_int],
        'sporco_amd_csc_ccmod_setcoef': [vp, ctypes.c_int],
        'sporco_amd_csc_ccmod_grad': [vp, ctypes.c_int, dptr],
        'sporco_amd_csc_ccmod_eval': [vp, ctypes.c_int, dptr],
        'sporco_amd_csc_ccmod_prox_step': [vp, dbl, i32, i32, i32],
        'sporco_amd_csc_ccmod_cnstr': [vp, i32, i32, i32, dptr],
        'sporco_amd_csc_ccmod_getdict': [vp, i32, i32, vp],
        'sporco_amd_csc_cns_init': [vp, vp, dbl],
        'sporco_amd_csc_cns_md_init': [vp, vp],
        'sporco_amd_csc_cns_mean_ptr': [vp, ctypes.POINTER(vp), ctypes.POINTER(i64)],
        'sporco_amd_csc_set_data_mask': [vp, vp, ctypes.POINTER(i64)],
        'sporco_amd_csc_masked_grad': [vp, ctypes.c_int, i32, i32, dptr],
        'sporco_amd_csc_cns_iter': [vp, ctypes.POINTER(CnsParams), dptr],
        'sporco_amd_csc_ccmod_sgd_step': [vp, dbl, i32, i32, i32, dptr],
        'sporco_amd_csc_mdcpl_init': [vp, vp],
        'sporco_amd_csc_mdcpl_iter': [vp, ctypes.POINTER(AdmmParams), dptr],
        'sporco_amd_csc_dstep_init': [vp, vp],
        'sporco_amd_csc_dstep_md_init': [vp, vp, vp],
        'sporco_amd_csc_dstep_iter': [vp, ctypes.POINTER(DstepParams), dptr],
        'sporco_amd_csc_setdict_from_dstep': [vp, i32, i32],
        'sporco_amd_csc_asum': [vp, ctypes.c_int, dptr],
        'sporco_amd_csc_profile': [vp, ctypes.c_int],
        'sporco_amd_csc_profile_read': [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                        dptr, ctypes.POINTER(i64)],
        'sporco_amd_profile_slots': [],
        'sporco_amd_rfftn2': [ctypes.c_int, i32, i32, i64, vp, vp],
        'sporco_amd_irfftn2': [ctypes.c_int, i32, i32, i64, vp, vp],
        'sporco_amd_solvedbi_sm': [ctypes.c_int, i64, i64, i32, vp, dbl, vp, vp],
        'sporco_amd_inner': [ctypes.c_int, i64, i64, i32, vp, vp, vp],
        'sporco_amd_dev_malloc': [ctypes.c_size_t, ctypes.POINTER(vp)],
        'sporco_amd_dev_free': [vp],
        'sporco_amd_dev_upload': [vp, vp, ctypes.c_size_t],
        'sporco_amd_dev_download': [vp, vp, ctypes.c_size_t],
        'sporco_amd_dev_axpby': [ctypes.c_int, i64, dbl, vp, dbl, vp, vp],
        'sporco_amd_tikhonov_filter_dev': [ctypes.c_int, ctypes.c_int32, ctypes.c_int32, i64, vp, dbl,
                                           ctypes.c_int32, vp, vp],
        'sporco_amd_fftconv_dev': [ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(i64),
                                   vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(i64), vp,
                                   ctypes.c_int32, ctypes.c_int32, vp],
        'sporco_amd_csc_set_signal_dev': [vp, vp],
        'sporco_amd_csc_reconstruct_dev': [vp, ctypes.c_int, vp],
        'sporco_amd_transfer_stats': [ctypes.POINTER(i64), ctypes.c_int],
        'sporco_amd_comm_unique_id': [vp],
        'sporco_amd_comm_create': [vp, i32, i32, i32, ctypes.POINTER(vp)],
        'sporco_amd_comm_destroy': [vp],
        'sporco_amd_comm_info': [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)],
        'sporco_amd_comm_allreduce': [vp, vp, i64, ctypes.c_int, ctypes.c_int, vp],
        'sporco_amd_comm_allreduce_host': [vp, dptr, i32, ctypes.c_int],
        'sporco_amd_csc_set_comm': [vp, vp],
        'sporco_amd_prox_l1': [ctypes.c_int, i64, vp, dbl, vp],
        'sporco_amd_prox_l1w': [ctypes.c_int, ctypes.POINTER(i64), vp, ctypes.POINTER(i64), vp, vp],
        'sporco_amd_prox_sl1l2': [ctypes.c_int, i64, i32, i64, vp, dbl, dbl, vp],
        'sporco_amd_rfl2norm2': [ctypes.c_int, i32, i32, i64, vp, dptr],
    }
    for name, argtypes in sig.items():
        getattr(lib, name).argtypes = argtypes
    _lib, _lib_path = lib, path
    return lib


def lib():
    """Return the loaded library, loading the default one on first use."""
    if _lib is None:
        load()
    return _lib


def library_path():
    return _lib_path


def transfer_stats(reset=False):
    """Host <-> device traffic of the library since the last reset:
    ``{'h2d_bytes', 'h2d_calls', 'd2h_bytes', 'd2h_calls'}``."""
    out = (ctypes.c_int64 * 4)()
    check(lib().sporco_amd_transfer_stats(out, 1 if reset else 0))
    return dict(h2d_bytes=out[0], h2d_calls=out[1], d2h_bytes=out[2], d2h_calls=out[3])


def check(rc):
    if rc != 0:
        msg = lib().sporco_amd_last_error()
        raise BackendError("sporco_amd backend error %d: %s" %
                           (rc, msg.decode('utf-8', 'replace') if msg else '?'))


def device_count():
    n = ctypes.c_int(0)
    check(lib().sporco_amd_device_count(ctypes.byref(n)))
    return n.value


def device_info(device=0):
    name = ctypes.create_string_buffer(256)
    cu = ctypes.c_int(0)
    mem = ctypes.c_size_t(0)
    check(lib().sporco_amd_device_info(device, name, 256, ctypes.byref(cu),
                                       ctypes.byref(mem)))
    return name.value.decode(), cu.value, mem.value


def dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return F32
    if dtype == np.float64:
        return F64
    raise TypeError("sporco_amd supports float32 and float64 data, not %s" % dtype)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _carr(a, dtype):
    a = np.asarray(a)
    if np.iscomplexobj(a) and not np.issubdtype(np.dtype(dtype), np.complexfloating):
        # (the reference solves complex-valued problems with complex FFTs,
        # sporco/admm/cbpdn.py:213-217; this backend's transforms are real-to-complex)
        raise NotImplementedError("complex-valued signals, dictionaries and weights are not "
                                  "supported by the sporco_amd backend")
    return np.ascontiguousarray(a, dtype=dtype)


class Solver(object):
    """Owner of one device-side ConvBPDN problem (opaque C handle)."""

    def __init__(self, H, W, C, N, K, dtype, device=0, stream=None, Cd=1, depth=1):
        """``C``: channels of the signal.  ``Cd`` > 1 (== C): multi-channel dictionary, the
        coefficient arrays then have a single channel (cnvrep.py:186-194).  ``depth`` > 1: a
        volume handle (dimN = 3), ``H`` = depth * height (include/sporco_amd.h
        sporco_amd_csc_create_volume)."""
        self.Cd = int(Cd)
        self.depth = int(depth)
        self.Cs = int(C)
        self.dims = (int(H), int(W), 1 if self.Cd > 1 else int(C), int(N), int(K))
        self.dtype = np.dtype(dtype)
        self.cdtype = np.dtype(np.complex64 if self.dtype == np.float32
                               else np.complex128)
        d = Dims(int(H), int(W), int(C), int(N), int(K), dtype_code(dtype))
        h = ctypes.c_void_p()
        if self.depth > 1:
            check(lib().sporco_amd_csc_create_volume(ctypes.byref(d), self.depth, int(device),
                                                     ctypes.c_void_p(stream or 0), ctypes.byref(h)))
        else:
            check(lib().sporco_amd_csc_create_mc(ctypes.byref(d), self.Cd, int(device),
                                                 ctypes.c_void_p(stream or 0),
                                                 ctypes.byref(h)))
        self._h = h
        self._lib = lib()

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.sporco_amd_csc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- shapes -----------------------------------------------------------
    @property
    def shape_x(self):
        return self.dims

    @property
    def shape_xf(self):
        H, W, C, N, K = self.dims
        return (H, W // 2 + 1, C, N, K)

    def var_shape_dtype(self, var):
        H, W, C, N, K = self.dims
        Wf = W // 2 + 1
        if var == VAR_DF:
            return (H, Wf, self.Cd, 1, K), self.cdtype
        if var == VAR_SF:
            return (H, Wf, self.Cs, N, 1), self.cdtype
        if var in (VAR_MY0, VAR_MU0, VAR_DMY0, VAR_DMU0):
            return (H, W, self.Cs, N, 1), self.dtype
        if var in (VAR_XF, VAR_YF, VAR_XFPRV, VAR_YFPRV, VAR_VF, VAR_GF, VAR_T0, VAR_T1, VAR_T2,
                   VAR_ZF):
            return (H, Wf, C, N, K), self.cdtype
        if var in (VAR_DX, VAR_DSX, VAR_DSU):
            return (H, W, self.Cd, 1, K), self.dtype
        if VAR_DXF <= var <= VAR_DT2:
            return (H, Wf, self.Cd, 1, K), self.cdtype
        if var in (VAR_CX, VAR_CU) and self.Cd > 1:
            # consensus copies of a multi-channel dictionary: one (Cd, K) block per image
            return (H, W, N, self.Cd, K), self.dtype
        return (H, W, C, N, K), self.dtype

    # -- set-up -----------------------------------------------------------
    def sync(self):
        check(self._lib.sporco_amd_csc_sync(self._h))

    def stream_handle(self):
        """hipStream_t (as an int) of every launch this handle makes."""
        out = ctypes.c_void_p(0)
        check(self._lib.sporco_amd_csc_stream(self._h, ctypes.byref(out)))
        return int(out.value or 0)

    def set_hint(self, what, value):
        check(self._lib.sporco_amd_csc_set_hint(self._h, int(what), int(value)))

    def set_comm(self, comm_handle):
        """Attach a native RCCL communicator (sporco_amd_csc_set_comm; None detaches): the
        device-driven solve then all-reduces its per-iteration sums itself."""
        check(self._lib.sporco_amd_csc_set_comm(self._h, ctypes.c_void_p(comm_handle or 0)))

    def query(self, what):
        out = ctypes.c_int(0)
        check(self._lib.sporco_amd_csc_query(self._h, int(what), ctypes.byref(out)))
        return out.value

    def placement_report(self):
        """Where the handle put the arrays one kernel writes at the same time
        (sporco_amd_csc_placement_report): a list of dicts, one per decision."""
        import json
        buf = ctypes.create_string_buffer(8192)
        check(self._lib.sporco_amd_csc_placement_report(self._h, buf, len(buf)))
        return json.loads(buf.value.decode() or '[]')

    def _query(self, what):
        return bool(self.query(what))

    def uses_fused_cols(self):
        """True when the register-resident column kernel serves this shape."""
        return self._query(0)

    def uses_fused_rows(self):
        """True when an ADMM iteration runs as the three fused launches."""
        return self._query(1)

    def uses_fused_pgm(self):
        """True when pgm_iter / the tile-major D-step serve this shape."""
        return self._query(2)

    def set_signal(self, S):
        H, W, C, N, K = self.dims
        S = _carr(S, self.dtype).reshape(H, W, self.Cs, N)
        check(self._lib.sporco_amd_csc_set_signal(self._h, _ptr(S)))

    def set_dict(self, D):
        K = self.dims[4]
        D = _carr(D, self.dtype)
        dH, dW = D.shape[0], D.shape[1]
        D = D.reshape(dH, dW, self.Cd * K)      # (dH, dW, Cd, 1, K) is contiguous as (.., Cd K)
        check(self._lib.sporco_amd_csc_set_dict(self._h, _ptr(D), dH, dW))

    def set_dict_imag(self, D_imag):
        """The imaginary part of a complex dictionary (complex mode, sporco_amd_csc_set_dict_imag);
        None: back to a real dictionary."""
        if D_imag is None:
            check(self._lib.sporco_amd_csc_set_dict_imag(self._h, None, 1, 1))
            return
        K = self.dims[4]
        D = _carr(D_imag, self.dtype)
        dH, dW = D.shape[0], D.shape[1]
        check(self._lib.sporco_amd_csc_set_dict_imag(self._h, _ptr(D.reshape(dH, dW, K)), dH, dW))

    def _set_weight(self, fn, w):
        if w is None:
            check(fn(self._h, None, None))
            return
        w = _carr(w, self.dtype)
        shape = (ctypes.c_int64 * 5)(*w.shape)
        check(fn(self._h, _ptr(w), shape))

    def set_l1_weight(self, w):
        self._set_weight(self._lib.sporco_amd_csc_set_l1_weight, w)

    def set_l21_weight(self, w):
        self._set_weight(self._lib.sporco_amd_csc_set_l21_weight, w)

    def set_filter_sizes(self, sizes):
        """Per-filter supports [(rows, cols)] * K of a multi-scale dictionary, or None.
        (The call synchronises the stream and re-allocates two small device arrays: unchanged
        sizes -- an online learner sets them every step -- return at once.)"""
        key = None if sizes is None else tuple((int(s[0]), int(s[1])) for s in sizes)
        if getattr(self, '_filter_sizes_set', False) and getattr(self, '_filter_sizes', None) == key:
            return
        self._filter_sizes, self._filter_sizes_set = key, True
        if sizes is None:
            check(self._lib.sporco_amd_csc_set_filter_sizes(self._h, None, None))
            return
        H, W, C, N, K = self.dims
        if len(sizes) != K:
            self._filter_sizes_set = False
            raise ValueError("one (rows, cols) pair per filter")
        fh = (ctypes.c_int32 * K)(*[int(s[0]) for s in sizes])
        fw = (ctypes.c_int32 * K)(*[int(s[1]) for s in sizes])
        check(self._lib.sporco_amd_csc_set_filter_sizes(self._h, fh, fw))

    def set_ams_mask(self, w):
        """AddMaskSim mask, 5-D with every axis 1 or full and a singleton filter axis."""
        self._set_weight(self._lib.sporco_amd_csc_set_ams_mask, w)

    def set_data_mask(self, w):
        """Data-fidelity mask of the *Mask PGM classes, 5-D, singleton filter axis."""
        self._set_weight(self._lib.sporco_amd_csc_set_data_mask, w)

    def masked_grad(self, var, dstep, write_grad):
        """Masked data-fidelity gradient / evaluation (sporco_amd_csc_masked_grad)."""
        out = self._out()
        check(self._lib.sporco_amd_csc_masked_grad(self._h, int(var), 1 if dstep else 0,
                                                   int(write_grad), out))
        return list(out)

    def set_grad_weight(self, w):
        """K per-filter weights of the gradient penalty, or None for 1."""
        if w is None:
            check(self._lib.sporco_amd_csc_set_grad_weight(self._h, None))
            return
        w = _carr(w, self.dtype).ravel()
        if w.size != self.dims[4]:
            raise ValueError("GradWeight must hold one value per filter")
        check(self._lib.sporco_amd_csc_set_grad_weight(self._h, _ptr(w)))

    # -- transfers --------------------------------------------------------
    def upload(self, var, a):
        shape, dt = self.var_shape_dtype(var)
        a = _carr(a, dt)
        if a.size != int(np.prod(shape)):
            raise ValueError("array of shape %s cannot fill state of shape %s" %
                             (a.shape, shape))
        check(self._lib.sporco_amd_csc_upload(self._h, var, _ptr(a)))

    def download(self, var):
        shape, dt = self.var_shape_dtype(var)
        out = np.empty(shape, dtype=dt)
        check(self._lib.sporco_amd_csc_download(self._h, var, _ptr(out)))
        return out

    def device_reals(self, var):
        """(number of real scalars, torch dtype) of the device array behind ``var`` -- the
        device layout, i.e. with the padding filter of an odd dictionary."""
        import torch
        shape, dt = self.var_shape_dtype(var)
        kdev = self.query(QUERY_DEVICE_FILTERS)
        n = int(np.prod(shape[:-1])) * (kdev if shape[-1] == self.dims[4] else shape[-1])
        if np.dtype(dt).kind == 'c':
            n *= 2
        return n, (torch.float32 if self.dtype == np.float32 else torch.float64)

    def device_ptr(self, var):
        p = ctypes.c_void_p()
        check(self._lib.sporco_amd_csc_device_ptr(self._h, var, ctypes.byref(p)))
        return p.value

    # -- ADMM ---------------------------------------------------------------
    @staticmethod
    def _out():
        return (ctypes.c_double * OUT_COUNT)()

    def admm_iter(self, params):
        out = self._out()
        check(self._lib.sporco_amd_csc_admm_iter(self._h, ctypes.byref(params), out))
        return list(out)

    def admm_run(self, params, ctrl, reduce=None):
        """Device-driven solve (sporco_amd_csc_admm_run): up to ``ctrl.max_iter`` iterations,
        stopping on the tolerance test evaluated on the device.  ``reduce(sums_dev_ptr)``, when
        given, sums the 16 doubles at that device address over the ranks.  Returns ``(records,
        rho, u_scale)``, or None when this configuration has to be driven from the host."""
        n = ctypes.c_int32(0)
        rho, usc = ctypes.c_double(0.0), ctypes.c_double(0.0)
        recs = (AdmmRecord * max(int(ctrl.max_iter), 1))()
        cb = REDUCE_FN(lambda user, ptr: reduce(ptr)) if reduce is not None else \
            ctypes.cast(None, REDUCE_FN)
        rc = self._lib.sporco_amd_csc_admm_run(self._h, ctypes.byref(params), ctypes.byref(ctrl),
                                               recs, ctypes.byref(n), ctypes.byref(rho),
                                               ctypes.byref(usc), cb, None)
        if rc == EUNSUPPORTED:
            return None
        check(rc)
        return [recs[i] for i in range(n.value)], rho.value, usc.value

    def mdcpl_init(self, S):
        """ConvBPDNMaskDcpl state: keeps the real signal on the device, zeroes Y and U."""
        H, W, C, N, K = self.dims
        S = _carr(S, self.dtype)
        if S.size != H * W * self.Cs * N:
            raise ValueError("signal of shape %s does not match the solver" % (S.shape,))
        check(self._lib.sporco_amd_csc_mdcpl_init(self._h, _ptr(S)))

    def mdcpl_iter(self, params):
        out = self._out()
        check(self._lib.sporco_amd_csc_mdcpl_iter(self._h, ctypes.byref(params), out))
        return list(out)

    def admm_iter_dev(self, params, out_dev_ptr):
        check(self._lib.sporco_amd_csc_admm_iter_dev(self._h, ctypes.byref(params),
                                                     ctypes.c_void_p(out_dev_ptr)))

    def admm_xstep(self, params):
        out = self._out()
        check(self._lib.sporco_amd_csc_admm_xstep(self._h, ctypes.byref(params), out))
        return list(out)

    def admm_relax(self, rlx):
        check(self._lib.sporco_amd_csc_admm_relax(self._h, float(rlx)))

    def admm_ystep(self, params):
        check(self._lib.sporco_amd_csc_admm_ystep(self._h, ctypes.byref(params)))

    def admm_ustep(self, params):
        check(self._lib.sporco_amd_csc_admm_ustep(self._h, ctypes.byref(params)))

    def admm_stats(self, params):
        out = self._out()
        check(self._lib.sporco_amd_csc_admm_stats(self._h, ctypes.byref(params), out))
        return list(out)

    def scale_u(self, s):
        check(self._lib.sporco_amd_csc_scale_u(self._h, float(s)))

    def reconstruct(self, var):
        H, W, C, N, K = self.dims
        out = np.empty((H, W, self.Cs, N, 1), dtype=self.dtype)
        check(self._lib.sporco_amd_csc_reconstruct(self._h, var, _ptr(out)))
        return out

    def reconstruct_dev(self, var, dst_ptr):
        """The same into a device array (H, W, Cs, N) at ``dst_ptr``."""
        check(self._lib.sporco_amd_csc_reconstruct_dev(self._h, var, ctypes.c_void_p(dst_ptr)))

    def set_signal_dev(self, ptr):
        """set_signal from a device array (H, W, Cs, N): no host copy."""
        check(self._lib.sporco_amd_csc_set_signal_dev(self._h, ctypes.c_void_p(ptr)))

    def dhs_absmax(self):
        v = ctypes.c_double(0.0)
        check(self._lib.sporco_amd_csc_dhs_absmax(self._h, ctypes.byref(v)))
        return v.value

    # -- PGM ----------------------------------------------------------------
    def pgm_grad(self, var):
        out = self._out()
        check(self._lib.sporco_amd_csc_pgm_grad(self._h, var, out))
        return list(out)

    def pgm_iter(self, L, lmbda, beta, flags, dH, dW, want_stats, hold=False):
        """One fused default-option FISTA iteration (sporco_amd_csc_pgm_iter).  hold: a
        backtracking trial -- out[PGM_LIN], out[PGM_DXY2] too, and the new iterates wait for
        pgm_commit (the call may be repeated with another L)."""
        # (hold = 2: a trial of a rule that forms Yf itself -- no momentum output; after the
        # commit VAR_YF is the previous iteration's Yf and VAR_YFPRV the one this trial used)
        p = PgmParams(float(L), float(lmbda), float(beta), int(flags), int(dH), int(dW),
                      1 if want_stats else 0, int(hold))
        out = self._out()
        check(self._lib.sporco_amd_csc_pgm_iter(self._h, ctypes.byref(p), out))
        return list(out)

    def pgm_commit(self):
        check(self._lib.sporco_amd_csc_pgm_commit(self._h))

    def cns_init(self, Y0, rho):
        """Consensus D-step state: Y = Y0 (H, W, 1, 1, K) or zero, U_n = Y0 / rho or zero."""
        if Y0 is None:
            check(self._lib.sporco_amd_csc_cns_init(self._h, None, float(rho)))
            return
        H, W, C, N, K = self.dims
        Y0 = _carr(Y0, self.dtype).reshape(H, W, self.Cd, K)
        check(self._lib.sporco_amd_csc_cns_init(self._h, _ptr(Y0), float(rho)))

    def cns_md_init(self, S):
        """State of the masked consensus D-step: the real signal on the device, Y1 = U1 = 0."""
        H, W, C, N, K = self.dims
        S = _carr(S, self.dtype)
        if S.size != H * W * self.Cs * N:
            raise ValueError("signal of shape %s does not match the solver" % (S.shape,))
        check(self._lib.sporco_amd_csc_cns_md_init(self._h, _ptr(S)))

    def cns_mean_ptr(self):
        """(device address, element count) of the consensus-mean buffer (phase 1 -> phase 2)."""
        p, n = ctypes.c_void_p(), ctypes.c_int64(0)
        check(self._lib.sporco_amd_csc_cns_mean_ptr(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def cns_iter(self, rho, rlx, u_scale, flags, dH, dW, zero_mean, mask_dcpl=False, phase=0):
        """One consensus D-step iteration (sporco_amd_csc_cns_iter), or one of its two phases
        when the images are sharded over ranks."""
        p = CnsParams(float(rho), float(rlx), float(u_scale), int(flags), int(dH), int(dW),
                      1 if zero_mean else 0, 1 if mask_dcpl else 0, int(phase))
        out = self._out()
        check(self._lib.sporco_amd_csc_cns_iter(self._h, ctypes.byref(p), out))
        return list(out)

    def ccmod_sgd_step(self, eta, dH, dW, zero_mean):
        """Projected SGD step on the X-step dictionary (sporco_amd_csc_ccmod_sgd_step)."""
        out = self._out()
        check(self._lib.sporco_amd_csc_ccmod_sgd_step(self._h, float(eta), int(dH), int(dW),
                                                      1 if zero_mean else 0, out))
        return list(out)

    def dstep_init(self, Y0):
        """Single-copy ADMM D-step state: Y = U = Y0 (H, W, Cd, 1, K) or zero, Xf = 0."""
        if Y0 is None:
            check(self._lib.sporco_amd_csc_dstep_init(self._h, None))
            return
        H, W, C, N, K = self.dims
        Y0 = _carr(Y0, self.dtype).reshape(H, W, self.Cd, K)
        check(self._lib.sporco_amd_csc_dstep_init(self._h, _ptr(Y0)))

    def dstep_md_init(self, Y0, S):
        """Mask-decoupling D-step state: dstep_init(Y0), block 0 zeroed, real signal kept."""
        H, W, C, N, K = self.dims
        y0 = None if Y0 is None else _carr(Y0, self.dtype).reshape(H, W, self.Cd, K)
        s = None if S is None else _carr(S, self.dtype)
        check(self._lib.sporco_amd_csc_dstep_md_init(self._h, None if y0 is None else _ptr(y0),
                                                     None if s is None else _ptr(s)))

    def dstep_iter(self, method, rho, rlx, u_scale, flags, dH, dW, zero_mean, cg_tol=1e-3,
                   cg_maxiter=1000, mask_dcpl=False):
        """One IterSM / CG D-step iteration (sporco_amd_csc_dstep_iter)."""
        p = DstepParams(float(rho), float(rlx), float(u_scale), float(cg_tol), int(flags), int(dH),
                        int(dW), 1 if zero_mean else 0, int(method), int(cg_maxiter),
                        1 if mask_dcpl else 0)
        out = self._out()
        check(self._lib.sporco_amd_csc_dstep_iter(self._h, ctypes.byref(p), out))
        return list(out)

    def pgm_eval(self, var):
        out = self._out()
        check(self._lib.sporco_amd_csc_pgm_eval(self._h, var, out))
        return list(out)

    def pgm_prox_step(self, L, lmbda, flags, dH, dW):
        out = self._out()
        check(self._lib.sporco_amd_csc_pgm_prox_step(self._h, float(L), float(lmbda),
                                                     int(flags), int(dH), int(dW), out))
        return list(out)

    def lincomb(self, dst, a, va, b=0.0, vb=-1, c=0.0, vc=-1):
        check(self._lib.sporco_amd_csc_lincomb(self._h, dst, float(a), va, float(b), vb,
                                               float(c), vc))

    def pair_stats(self, va, vb=-1, vg=-1):
        out = self._out()
        check(self._lib.sporco_amd_csc_pair_stats(self._h, va, vb, vg, out))
        return list(out)[:4]

    def pgm_resid(self, var, slot):
        """e = sum_m Df var - Sf into residual slot 0..3 (sporco_amd_csc_pgm_resid)."""
        check(self._lib.sporco_amd_csc_pgm_resid(self._h, var, slot))

    def pgm_resid_stats(self, a, b=-1, c=-1, d=-1):
        """(<g1, g1>, <g1, H g1>, <dx, dg>) from residual slots (sporco_amd_csc_pgm_resid_stats)."""
        out = self._out()
        check(self._lib.sporco_amd_csc_pgm_resid_stats(self._h, a, b, c, d, out))
        return list(out)[:3]

    def copy(self, dst, src):
        check(self._lib.sporco_amd_csc_copy(self._h, dst, src))

    # -- dictionary update ---------------------------------------------------
    def ccmod_setcoef(self, var):
        check(self._lib.sporco_amd_csc_ccmod_setcoef(self._h, var))

    def ccmod_grad(self, var):
        out = self._out()
        check(self._lib.sporco_amd_csc_ccmod_grad(self._h, var, out))
        return list(out)

    def ccmod_eval(self, var):
        out = self._out()
        check(self._lib.sporco_amd_csc_ccmod_eval(self._h, var, out))
        return list(out)

    def ccmod_prox_step(self, L, dH, dW, zero_mean):
        check(self._lib.sporco_amd_csc_ccmod_prox_step(self._h, float(L), int(dH), int(dW),
                                                       1 if zero_mean else 0))

    def ccmod_cnstr(self, dH, dW, zero_mean):
        out = self._out()
        check(self._lib.sporco_amd_csc_ccmod_cnstr(self._h, int(dH), int(dW),
                                                   1 if zero_mean else 0, out))
        return out[0]

    def ccmod_getdict(self, dH, dW):
        K = self.dims[4]
        out = np.empty((dH, dW, self.Cd, 1, K), dtype=self.dtype)
        check(self._lib.sporco_amd_csc_ccmod_getdict(self._h, int(dH), int(dW), _ptr(out)))
        return out

    def setdict_from_dstep(self, dH, dW):
        check(self._lib.sporco_amd_csc_setdict_from_dstep(self._h, int(dH), int(dW)))

    def asum(self, var):
        out = self._out()
        check(self._lib.sporco_amd_csc_asum(self._h, var, out))
        return out[0]

    def fft_var(self, real_var, cplx_var):
        check(self._lib.sporco_amd_csc_fft_var(self._h, real_var, cplx_var))

    def ifft_var(self, cplx_var, real_var):
        check(self._lib.sporco_amd_csc_ifft_var(self._h, cplx_var, real_var))

    # -- timing -------------------------------------------------------------
    def profile(self, enable):
        check(self._lib.sporco_amd_csc_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        """Return {kernel name: (total_ms, launches)} and reset the counters."""
        res = {}
        for slot in range(self._lib.sporco_amd_profile_slots()):
            name = ctypes.c_char_p()
            ms = ctypes.c_double(0.0)
            cnt = ctypes.c_int64(0)
            check(self._lib.sporco_amd_csc_profile_read(self._h, slot, ctypes.byref(name),
                                                        ctypes.byref(ms), ctypes.byref(cnt)))
            res[name.value.decode()] = (ms.value, cnt.value)
        return res
