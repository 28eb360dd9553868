#!/usr/bin/env python3
"""bench.py -- ConvBPDN ADMM iterations/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 ...      (starts its own ranks through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ``admm.cbpdn.ConvBPDN`` on 512x512
greyscale images, K = 64 filters of 8x8, N = 32 images PER GPU, float32,
default options (relaxation 1.8, adaptive rho every iteration, residuals and
objective evaluated every iteration) -- i.e. what a reference user gets from
``ConvBPDN(D, S, lmbda).solve()``.  Synthetic data: randn images, l2-normalised
randn filters, RandomState(12345 + rank), lambda = 0.05.
Inputs are resident in HBM before the timed region (the solver object owns
them); the timed region is ``solve()`` for exactly K iterations.

Multi-GPU: images shard over ranks (weak scaling: 32 images per GPU, so 8 GPUs
run the N = 256 problem); the only communication is one all-reduce (RCCL) of
the 16 per-iteration scalars.  ``value`` sums the per-GPU iteration rates, i.e.
it is the number of 512x512x64x32-sized ADMM iterations completed per second by
the whole job.

One "step" = one ADMM iteration.  Rank 0 prints ONE JSON line.

The line's `configs` object carries the other BASELINE.json configurations as one GPU sees them
(config 1; config 3 as one of its 8 image shards; config 4 with and without BacktrackStandard;
config 5), each timed the same way, with a parity figure against a run of the UNMODIFIED reference
on the same kernel instantiations (committed fixtures, tests/golden/ -- oracle/make_golden.py is
the recipe) and the roofline of its dominant kernel.  `roofline.frac` is bytes really MOVED by
the dominant kernel (by construction: its inputs and outputs once; equal to the rocprofv3 PMC
traffic in profiles/hbm_traffic_bytes.json) / its HIP-event duration in this run / the 8 TB/s
HBM peak; the SURVEY.md 8(d) stage-bytes figure is kept beside it as `stage_model_ratio` -- stage
bytes / time / peak, which is NOT a distance to the peak: the single-array kernels move two thirds
of the bytes that model credits, so it exceeds 1 where they run near the roofline.
`cpu_baseline` times the unmodified reference (staged into the git-ignored oracle/_ref by
build()) in a subprocess on this box's host cores, and the NumPy port as a second leg.
`roofline.placement` lists where the library put the arrays the dominant kernel writes at the
same time (sporco_amd_csc_placement_report); `frac_check` states that no `frac`
of the line exceeds 1 (check_fracs walks the whole line before it is printed).
"""

import argparse
import gc
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)


def make_problem(H, W, K, N, rank, dtype=np.float32):
    rng = np.random.RandomState(12345 + rank)
    D = rng.randn(8, 8, K).astype(dtype)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, N).astype(dtype)
    return D, S


def make_problem_rgb(H, W, K, N, rank, dtype=np.float32):
    """Config 3's input: N RGB images (H, W, 3, N) and a single-channel dictionary (SURVEY.md 8(d):
    `ConvBPDNJoint(D[8,8,K], S[H,W,3,N], lambda=0.1, mu=0.01)`, Cd = 1, C = 3)."""
    rng = np.random.RandomState(12345 + rank)
    D = rng.randn(8, 8, K).astype(dtype)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, 3, N).astype(dtype)
    return D, S


def make_structured_problem(H, W, K, N, rank, dtype=np.float32, density=0.0027):
    """Sparse-synthesis input for the time-to-tolerance measurement (the recipe of the
    reference's known-answer test, tests/admm/test_cbpdn.py:160-165, at image size):
    S_n = sum_k d_k * x0_{n,k} with x0 sparse (P(|randn| > 3) = 0.27 % of the entries nonzero,
    standard normal values), circular convolution evaluated in float64."""
    import scipy.fft as sfft
    rng = np.random.RandomState(54321 + rank)
    D = rng.randn(8, 8, K)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    Df = sfft.rfft2(D, (H, W), axes=(0, 1), workers=-1)
    S = np.empty((H, W, N), dtype=dtype)
    for n in range(N):
        m = rng.rand(H, W, K) < density
        X0 = np.zeros((H, W, K))
        X0[m] = rng.randn(int(m.sum()))
        S[:, :, n] = sfft.irfft2(np.sum(Df * sfft.rfft2(X0, axes=(0, 1), workers=-1), axis=2),
                                 (H, W), axes=(0, 1), workers=-1)
    return D.astype(dtype), S


def byte_model(H, W, C, N, K, itemsize=4, grad_groups=0):
    """HBM bytes per launch of every profiled kernel slot (sporco_amd_csc_profile_read names), for
    arrays of P = C*N*K lines.  Returns (moved, algorithmic):

    * `moved` -- what the kernel moves BY CONSTRUCTION: each X-sized input and output once (the
      broadcast operands Df, Sf, denominators, weights -- 1/N or 1/K of an array -- are left out);
      the rocprofv3 PMC passes measure exactly these (profiles/hbm_traffic_bytes.json).
    * `algorithmic` -- SURVEY.md 8(d): each logical stage reading its inputs and writing its
      outputs once.  Differs from `moved` only for the kernels of the single-array state
      (DESIGN.md 4.1c), which read V where 8(d)'s stage reads Y and U and write V' for Y', U'."""
    P = C * N * K
    E = H * W * P * itemsize                   # one pass over a real X-sized array
    EF = H * (W // 2 + 1) * P * 2 * itemsize   # one pass over a half-spectrum array
    moved = {
        'rows_fwd': 2 * E + EF,                # read Y, U; write tile-major row spectra
        'rows_fwd_v': E + EF,                  # read V
        'fused_cols_sm': 2 * EF,               # column FFT + solve + IFFT in place (K > 64: slabs)
        'rows_inv_post': EF + 4 * E,           # read spectra, Y, U; write Y', U'
        'rows_inv_post_emit': 2 * EF + 4 * E,  # ... and the next iteration's row spectra
        'rows_inv_post_v': EF + 2 * E,         # read spectra, V; write V'
        'rows_inv_post_v_emit': 2 * EF + 2 * E,
        'fft_r2c_rows': 2 * E + EF, 'fft_c2c_cols_fwd': 2 * EF, 'sm_solve': 2 * EF,
        'fft_c2c_cols_inv': 2 * EF, 'fft_c2r_rows': EF + E, 'admm_post': 5 * E,
        # FISTA: spectra stay tile-major, X is rebuilt on demand
        'pgm_grad_ifft': 2 * EF,               # Yf -> gradient step -> column IFFT
        'pgm_rows_prox': 2 * EF,               # row IFFT -> prox -> row FFT
        'pgm_fft_momentum': 5 * EF,            # column FFT; read Xf, Yf; write Xf', Yf'
        # dictionary update
        'setcoef_rows': E + EF, 'setcoef_cols': 2 * EF,
        # Zf once; the per-group partial gradients are written and summed (dictionary sized x G)
        'ccmod_grad_tiled': EF + 2 * grad_groups * H * (W // 2 + 1) * K * 2 * itemsize,
    }
    alg = dict(moved)
    alg.update({'rows_fwd_v': 2 * E + EF, 'rows_inv_post_v': EF + 4 * E,
                'rows_inv_post_v_emit': 2 * EF + 4 * E})
    return moved, alg


def kernel_roofline(prof, moved, alg):
    """Per-kernel table from the library's HIP-event timings {name: (total_ms, launches)}."""
    out = {}
    for k, v in prof.items():
        if v[1] > 0 and k in moved:
            ms = v[0] / v[1]
            out[k] = {'avg_ms': round(ms, 4), 'launches': int(v[1]),
                      'moved_bytes': moved[k], 'GBps': round(moved[k] / ms / 1e6, 1),
                      'frac': round(moved[k] / ms / 1e6 / HBM_PEAK_GBPS, 4),
                      'stage_model_ratio': round(alg[k] / ms / 1e6 / HBM_PEAK_GBPS, 4)}
    return out


def check_fracs(obj, path='line'):
    """No fraction of a peak above 1 may leave this program (a `frac` > 1 in a roofline table means
    the time it was formed from is not the time of the work it is credited with -- e.g. idle
    dispatches averaged in).  Returns the offending paths; main() refuses to print them."""
    bad = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            # (`stage_model_ratio` credits the stage bytes of SURVEY 8(d), of which the single-array
            # kernels move two thirds: it is named as what it is, not as a fraction of the peak)
            if k == 'frac' and isinstance(v, (int, float)) and v > 1.0:
                bad.append('%s.%s=%.3f' % (path, k, v))
            else:
                bad.extend(check_fracs(v, '%s.%s' % (path, k)))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            bad.extend(check_fracs(v, '%s[%d]' % (path, i)))
    return bad


def dominant(table):
    return max(table, key=lambda k: table[k]['avg_ms'] * table[k]['launches'])


def roofline_of(table, extra=None):
    """The `roofline` object of one configuration: its dominant kernel (largest total time in the
    profiled pass), bytes moved / HIP-event duration / 8 TB/s."""
    k = dominant(table)
    t = table[k]
    r = {'bound': 'hbm', 'kernel': k, 'moved_bytes': t['moved_bytes'], 'avg_kernel_ms': t['avg_ms'],
         'achieved': t['GBps'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': t['frac'],
         'stage_model_ratio': t['stage_model_ratio']}
    if extra:
        r.update(extra)
    return r


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def load_fixture(name):
    p = os.path.join(REPO, 'tests', 'golden', name + '.npz')
    if not os.path.exists(p):
        return None
    with np.load(p) as g:
        return {k: g[k] for k in g.files}


def fixture_parity(fix_name, coef, coef_key, its, fields, tol=1e-4, trace_tol=1e-3, extra=None):
    """Compare a run of this library (float32) with the float64 run of the UNMODIFIED reference
    on the same inputs (tests/golden/<fix_name>.npz, written by oracle/make_golden.py in the
    authoring container): strided subsample of the coefficient array and per-iteration traces."""
    g = load_fixture(fix_name)
    if g is None:
        return {'pass': None, 'note': 'fixture %s absent' % fix_name}
    rel = rel_l2(coef[::16, ::16], g[coef_key])
    tr = {f: rel_l2(getattr(its, f), g['it_' + f]) for f in fields if 'it_' + f in g}
    out = {'vs': 'unmodified reference, float64 run of the same float32 inputs '
                 '(tests/golden/%s.npz)' % fix_name,
           'rel_l2': rel, 'compared': '%s[::16, ::16] (%d values)' % (coef_key.split('_')[0],
                                                                   g[coef_key].size),
           'iters': int(len(g['it_Iter'])), 'trace_rel_err': tr,
           'pass': bool(rel <= tol and (not tr or max(tr.values()) <= trace_tol))}
    if extra:
        out.update(extra)
    return out


def timed_solve(b, dev, warmup, steps):
    import gc
    b.opt['MaxMainIter'] = max(warmup, 1)
    b.solve()
    b.opt['MaxMainIter'] = steps
    # (no cyclic-garbage collection inside the timed region: collecting the previous leg's solver
    # there frees gigabytes of device memory, a device-wide stall of tens of milliseconds --
    # one run of the r04z session timed config 5 at half its rate for it)
    gc.collect()
    gc.disable()
    try:
        dev.sync()
        t0 = time.perf_counter()
        b.solve()
        dev.sync()
        return time.perf_counter() - t0
    finally:
        gc.enable()


def profiled_pass(b, dev, steps, host_loop=True):
    """Per-kernel HIP-event durations over `steps` more iterations of the same solver (the
    host-driven loop launches exactly the kernels that execute; see main())."""
    b.opt['MaxMainIter'] = steps
    dev.profile(True)
    if host_loop:
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
    try:
        b.solve()
    finally:
        os.environ.pop('SPORCO_AMD_HOST_LOOP', None)
    dev.sync()
    prof = dev.profile_read()
    dev.profile(False)
    return {k: v for k, v in prof.items() if v[1] > 0}


def iteration_summary(table, prof_steps, ms_per_step, alg_bytes_per_iter):
    mv = sum(t['moved_bytes'] * t['launches'] for t in table.values()) / float(prof_steps)
    return {'moved_bytes_per_iter': mv, 'frac': mv / ms_per_step / 1e6 / HBM_PEAK_GBPS,
            'algorithmic_bytes_per_iter': alg_bytes_per_iter,
            'stage_model_ratio': alg_bytes_per_iter / ms_per_step / 1e6 / HBM_PEAK_GBPS,
            'note': 'moved = sum over the profiled kernels of bytes moved by construction, per '
                    'iteration of the profiled pass; ms = the timed region'}


# ---- the other BASELINE.json configurations, one GPU -------------------------------------------
# (`--tiny`: the same legs at sizes the CPU fiber simulator of the test-suite finishes in seconds
# -- exercises this file's control flow without a GPU; its numbers mean nothing and the fixture
# comparisons, which are for the full shapes, are skipped)
TINY_NOTE = {'pass': None, 'note': '--tiny run: shapes differ from the fixtures, parity skipped'}

def run_config1(device, tiny=False):
    """admm.cbpdn.ConvBPDN, 256x256, K=32, N=1 (BASELINE configs[0])."""
    from sporco_amd.admm import cbpdn
    H = W = 128 if tiny else 256
    K = 32
    D, S = make_problem(H, W, K, 1, 0)
    par = TINY_NOTE
    if not tiny:
        # parity: the fixture is this very problem, 20 iterations, default options
        b = cbpdn.ConvBPDN(D, S[:, :, 0], 0.05, cbpdn.ConvBPDN.Options(
            {'MaxMainIter': 20, 'RelStopTol': 0.0}), dimK=0, device=device)
        Y = b.solve()
        par = fixture_parity('admm_config1_f64', Y.reshape(H, W, 1, 1, K), 'Y_sub', b.getitstat(),
                             ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'))
        del b
    b = cbpdn.ConvBPDN(D, S[:, :, 0], 0.05, cbpdn.ConvBPDN.Options(
        {'MaxMainIter': 3, 'RelStopTol': 0.0}), dimK=0, device=device)
    b._return_min = False
    steps, warm = (6, 2) if tiny else (500, 20)
    el = timed_solve(b, b._dev, warm, steps)
    ms = 1e3 * el / steps
    psteps = 4 if tiny else 50
    prof = profiled_pass(b, b._dev, psteps)
    moved, alg = byte_model(H, W, 1, 1, K)
    tab = kernel_roofline(prof, moved, alg)
    out = {'workload': 'admm.cbpdn.ConvBPDN %dx%d greyscale, K=%d 8x8 filters, N=1, lambda=0.05, '
                       'default options (BASELINE configs[0])' % (H, W, K),
           'steps': steps, 'warmup': warm, 'ms_per_step': ms, 'value': steps / el,
           'unit': 'iterations/s', 'parity': par,
           'roofline': roofline_of(tab, {'note': '8.4 MB per array: one tile per CU and pass, the '
                                                 'iteration is tile latency + launches, not HBM'}),
           'iteration': iteration_summary(tab, psteps, ms, 40 * H * W * K),
           'kernels': tab}
    del b
    return out


def run_config3(device, n_shard=32, tiny=False):
    """ConvBPDNJoint 512x512 RGB, K=128, N=256 over 8 GPUs: the shard of one GPU (N=32)."""
    from sporco_amd.admm import cbpdn
    H = W = 128 if tiny else 512
    K, C = 128, 3
    lm, mu = 0.1, 0.01
    if tiny:
        n_shard = 1
    # parity on image 0 of the same input, the same kernel instantiations (H = W = 512, K = 128
    # slab column pass, joint row epilogue in the single-array form): 6 iterations
    par = TINY_NOTE
    if not tiny:
        D, S = make_problem_rgb(H, W, K, 1, 0)
        b = cbpdn.ConvBPDNJoint(D, S, lm, mu, cbpdn.ConvBPDNJoint.Options(
            {'MaxMainIter': 6, 'RelStopTol': 0.0}), device=device)
        engaged = bool(b._dev.uses_fused_rows() and b._dev.uses_fused_cols() and b._fused_ok())
        Y = b.solve()
        par = fixture_parity('admm_config3_joint_n1_f64', Y, 'Y_sub', b.getitstat(),
                             ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'PrimalRsdl', 'DualRsdl', 'Rho'),
                             extra={'fused_kernels_engaged': engaged})
        del b, Y
    D, S = make_problem_rgb(H, W, K, n_shard, 0)
    b = cbpdn.ConvBPDNJoint(D, S, lm, mu, cbpdn.ConvBPDNJoint.Options(
        {'MaxMainIter': 3, 'RelStopTol': 0.0}), device=device)
    b._return_min = False
    steps, warm = (4, 1) if tiny else (10, 3)
    el = timed_solve(b, b._dev, warm, steps)
    ms = 1e3 * el / steps
    nxt = 2 if tiny else 20
    b.opt['MaxMainIter'] = nxt
    b._dev.sync()
    t0 = time.perf_counter()
    b.solve()
    b._dev.sync()
    el2 = time.perf_counter() - t0
    prof = profiled_pass(b, b._dev, 6)
    moved, alg = byte_model(H, W, C, n_shard, K)
    tab = kernel_roofline(prof, moved, alg)
    E = H * W * C * n_shard * K
    out = {'workload': 'admm.cbpdn.ConvBPDNJoint %dx%d RGB (C=3), K=%d 8x8 filters, lambda=0.1, '
                       'mu=0.01, default options: N=%d images = one of the 8 shards of BASELINE '
                       'configs[2] (N=256), %.1f GB per X-sized array'
                       % (H, W, K, n_shard, 4 * E / 1e9),
           'steps': steps, 'warmup': warm, 'ms_per_step': ms, 'value': steps / el,
           'unit': 'iterations/s (per shard; 8 shards run concurrently, one all-reduce of 16 '
                   'doubles per iteration)',
           'next_steps': {'steps': nxt, 'value': nxt / el2, 'ms_per_step': 1e3 * el2 / nxt},
           'parity': par, 'roofline': roofline_of(tab),
           'iteration': iteration_summary(tab, 6, ms, 40 * E), 'kernels': tab,
           'placement': b._dev.placement_report()}
    del b
    return out


def run_config3_sharded(device, rank, world, reducer, stream, sync_all, allmax, n_shard=32, tiny=False):
    """BASELINE configs[2] as the job it is: ConvBPDNJoint 512x512 RGB, K=128, N = 32 images per
    rank (8 ranks: the N = 256 problem), images sharded, one all-reduce of the 16 sums per
    iteration inside the device-driven loop.  Every rank runs this; `allmax(x)` is the maximum of a
    host float over the ranks.  Returns the leg's entry (identical on every rank)."""
    from sporco_amd.admm import cbpdn
    H = W = 128 if tiny else 512
    K, C = (32 if tiny else 128), 3
    if tiny:
        n_shard = 1
    D, _ = make_problem_rgb(H, W, K, 1, 0)              # the dictionary is replicated
    _, S = make_problem_rgb(H, W, K, n_shard, rank)     # this rank's images
    b = cbpdn.ConvBPDNJoint(D, S, 0.1, 0.01, cbpdn.ConvBPDNJoint.Options(
        {'MaxMainIter': 3, 'RelStopTol': 0.0}), device=device, stream=stream, reducer=reducer)
    b._return_min = False
    steps, warm = (3, 1) if tiny else (10, 3)
    b.opt['MaxMainIter'] = warm
    b.solve()
    b.opt['MaxMainIter'] = steps
    gc.collect()
    gc.disable()
    try:
        sync_all(b)
        t0 = time.perf_counter()
        b.solve()
        sync_all(b)
        own = time.perf_counter() - t0
    finally:
        gc.enable()
    slow, fast = allmax(own), -allmax(-own)
    its = b.getitstat()
    E = H * W * C * n_shard * K
    out = {'workload': 'admm.cbpdn.ConvBPDNJoint %dx%d RGB (C=3), K=%d, lambda=0.1, mu=0.01, default '
                       'options: N=%d images per rank x %d ranks = N=%d (BASELINE configs[2] at 8 ranks), '
                       'images sharded, one all-reduce of 16 doubles per iteration'
                       % (H, W, K, n_shard, world, n_shard * world),
           'steps': steps, 'warmup': warm, 'ms_per_step': 1e3 * slow / steps,
           'value': steps / slow * world,
           'unit': 'shard-iterations/s summed over ranks (N=%d images each): weak scaling' % n_shard,
           'iterations_per_s_of_the_sharded_problem': steps / slow,
           'ms_per_step_fastest_rank': 1e3 * fast / steps,
           'fused_kernels_engaged': bool(b._dev.uses_fused_rows() and b._dev.uses_fused_cols()),
           'stage_model_ratio_per_gpu': 40 * E * (steps / slow) / 1e9 / HBM_PEAK_GBPS,
           'final_rho': float(its.Rho[-1]), 'final_primal_rsdl': float(its.PrimalRsdl[-1]),
           'reducer': type(reducer).__name__}
    del b
    return out


def run_config4(device, backtrack, tiny=False):
    """pgm.cbpdn.ConvBPDN (FISTA), 512x512, K=64, N=32, L=500 (BASELINE configs[3])."""
    from sporco_amd.pgm import cbpdn as pc
    from sporco_amd.pgm.backtrack import BacktrackStandard
    H = W = 128 if tiny else 512
    K, N = 64, (2 if tiny else 32)

    def opts(iters):
        o = {'MaxMainIter': iters, 'RelStopTol': 0.0, 'L': 500.0}
        if backtrack:
            o['Backtrack'] = BacktrackStandard()
        return pc.ConvBPDN.Options(o)

    par = TINY_NOTE
    if not tiny:
        D, S = make_problem(H, W, K, 2, 0)
        b = pc.ConvBPDN(D, S, 0.05, opts(8), device=device)
        engaged = bool(b.dev.uses_fused_rows() and b.dev.uses_fused_pgm() and b._fused_ok())
        X = b.solve()
        fields = ('ObjFun', 'DFid', 'RegL1', 'Rsdl', 'L') + (('F_Btrack', 'Q_Btrack', 'IterBTrack')
                                                            if backtrack else ())
        par = fixture_parity('pgm_config4_n2_bt_f64' if backtrack else 'pgm_config4_n2_f64',
                             X.reshape(H, W, 1, 2, K), 'X_sub', b.getitstat(), fields,
                             extra={'fused_kernels_engaged': engaged})
        del b, X
    D, S = make_problem(H, W, K, N, 0)
    b = pc.ConvBPDN(D, S, 0.05, opts(3), device=device)
    b._return_min = False
    steps, warm = (4, 1) if tiny else (20, 3)
    el = timed_solve(b, b.dev, warm, steps)
    ms = 1e3 * el / steps
    prof = profiled_pass(b, b.dev, 10, host_loop=False)
    moved, alg = byte_model(H, W, 1, N, K)
    tab = kernel_roofline(prof, moved, alg)
    out = {'workload': 'pgm.cbpdn.ConvBPDN (FISTA) %dx%d greyscale, K=%d, N=%d, lambda=0.05, '
                       'L=500%s (BASELINE configs[3])'
                       % (H, W, K, N, ', BacktrackStandard' if backtrack else ''),
           'steps': steps, 'warmup': warm, 'ms_per_step': ms, 'value': steps / el,
           'unit': 'iterations/s', 'parity': par, 'roofline': roofline_of(tab),
           'iteration': iteration_summary(tab, 10, ms, 40 * H * W * N * K), 'kernels': tab,
           'placement': b.dev.placement_report()}
    del b
    return out


def run_config5(device, tiny=False):
    """dictlrn.cbpdndl.ConvBPDNDictLearn, 256x256, K=64, N=64, admm X-step / pgm D-step
    (BASELINE configs[4]); metric = outer iterations/s."""
    from sporco_amd import _lib
    from sporco_amd.dictlrn import cbpdndl
    H = W = 128 if tiny else 256
    K, N = 64, (4 if tiny else 64)
    # parity: the reference's own float64 run of 4 outer iterations at N = 4 (same kernels)
    g = load_fixture('cbpdndl_config5_n4_f64')
    par = TINY_NOTE if tiny else {'pass': None, 'note': 'fixture absent'}
    if g is not None and not tiny:
        rng = np.random.RandomState(515)
        D0 = rng.randn(8, 8, 64).astype(np.float32)
        S4 = rng.randn(256, 256, 4).astype(np.float32)
        opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 4, 'AccurateDFid': True},
                                                xmethod='admm', dmethod='pgm')
        d = cbpdndl.ConvBPDNDictLearn(D0, S4, float(g['lmbda']), opt, xmethod='admm', dmethod='pgm',
                                      device=device)
        D1 = d.solve()
        X = d.getcoef()
        its = d.getitstat()
        tr = {f: rel_l2(getattr(its, f), g['it_' + f]) for f in its._fields
              if 'it_' + f in g and f not in ('Iter', 'Cnstr')}
        rd, rx = rel_l2(D1.squeeze(), g['D1'].squeeze()), rel_l2(X[::16, ::16], g['X_sub'])
        par = {'vs': 'unmodified reference, float64 run, 4 outer iterations at N=4 '
                     '(tests/golden/cbpdndl_config5_n4_f64.npz)',
               'rel_l2_dictionary': rd, 'rel_l2_coefficients_subsample': rx, 'trace_rel_err': tr,
               'pass': bool(rd <= 1e-4 and rx <= 1e-4 and max(tr.values()) <= 1e-3)}
        del d
    D, S = make_problem(H, W, K, N, 0)
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 3}, xmethod='admm', dmethod='pgm')
    d = cbpdndl.ConvBPDNDictLearn(D, S, 0.1, opt, xmethod='admm', dmethod='pgm', device=device)
    dev = d.xstep._dev
    steps, warm = (4, 1) if tiny else (20, 3)
    el = timed_solve(d, dev, warm, steps)
    ms = 1e3 * el / steps
    # (host-driven loop: it launches exactly the kernels that execute -- the device-driven X-step
    # enqueues both epilogue variants and lets the unneeded one return at once, and an average
    # over those idle dispatches is not a kernel time)
    prof = profiled_pass(d, dev, 10, host_loop=True)
    groups = dev.query(_lib.QUERY_CCMOD_GROUPS)      # ccmod_grad's image groups (api_dictupdate.inc)
    moved, alg = byte_model(H, W, 1, N, K, grad_groups=groups)
    # the generic FFT slots of this configuration are the D-step's transforms of the dictionary
    # (H, W, K): dictionary-sized, 1/N of the arrays above
    dm, da = byte_model(H, W, 1, 1, K)
    for k in ('fft_r2c_rows', 'fft_c2c_cols_fwd', 'fft_c2c_cols_inv', 'fft_c2r_rows'):
        moved[k], alg[k] = dm[k] - (H * W * K * 4 if k == 'fft_r2c_rows' else 0), da[k]
    tab = kernel_roofline(prof, moved, alg)
    out = {'workload': 'dictlrn.cbpdndl.ConvBPDNDictLearn %dx%d greyscale, K=%d 8x8 filters, N=%d, '
                       'lambda=0.1, xmethod admm / dmethod pgm, one inner iteration each, default '
                       'options (BASELINE configs[4])' % (H, W, K, N),
           'steps': steps, 'warmup': warm, 'ms_per_step': ms, 'value': steps / el,
           'unit': 'outer iterations/s', 'parity': par, 'roofline': roofline_of(tab),
           'iteration': iteration_summary(tab, 10, ms, 56 * H * W * N * K), 'kernels': tab,
           'placement': dev.placement_report()}
    del d
    return out


def run_other_configs(device, which, tiny=False):
    import gc
    legs = {'config1': lambda: run_config1(device, tiny=tiny),
            'config3_shard': lambda: run_config3(device, tiny=tiny),
            'config4': lambda: run_config4(device, False, tiny=tiny),
            'config4_backtrack': lambda: run_config4(device, True, tiny=tiny),
            'config5': lambda: run_config5(device, tiny=tiny)}
    out = {}
    for name, fn in legs.items():
        if which != 'all' and name.split('_')[0] not in which.split(','):
            continue
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:   # a failing side leg must not cost the headline line
            out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        out[name]['leg_seconds'] = round(time.perf_counter() - t0, 1)
        gc.collect()
    return out


def run_next_rows(device, tiny=False):
    """Short driver-timed legs for what is built beyond the BASELINE configurations (SURVEY 8(f)
    rows, PGM step-size rules, the generic chain every size / precision outside the register
    kernels runs): iterations/s with everything resident, 3 warm-up + 20 timed iterations each.
    Parity of these paths: tests/ (fixtures of the unmodified reference); here only the rate and
    which kernels served it."""
    import gc
    from sporco_amd.admm import cbpdn as ac
    from sporco_amd.pgm import cbpdn as pc
    from sporco_amd.pgm.backtrack import BacktrackRobust
    rng = np.random.RandomState(12345)
    H = 128 if tiny else 512
    K, N8 = (8, 2) if tiny else (64, 8)

    def rate(b, dev, iters=20):
        b._return_min = False
        return iters / timed_solve(b, dev, 3, iters)

    D = rng.randn(8, 8, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, H, N8).astype(np.float32)
    Wm = (rng.rand(H, H, N8) > 0.3).astype(np.float32)
    o0 = {'MaxMainIter': 3, 'RelStopTol': 0.0}

    def leg_maskdcpl():
        b = ac.ConvBPDNMaskDcpl(D, S, 0.05, Wm, ac.ConvBPDNMaskDcpl.Options(o0), device=device)
        return {'workload': 'admm.cbpdn.ConvBPDNMaskDcpl %dx%d K=%d N=%d f32' % (H, H, K, N8),
                'value': rate(b, b._dev), 'unit': 'iterations/s',
                'path': 'register-resident kernels' if b._dev.uses_fused_rows() else 'generic chain'}

    def leg_maskdcpl_mr():
        # mask decoupling at a mixed-radix shape (round 6), and the generic chain of the same library
        h, w = (48, 40) if tiny else (480, 320)
        r2 = np.random.RandomState(2)
        Sg = r2.randn(h, w, N8).astype(np.float32)
        Wg = (r2.rand(h, w, N8) > 0.3).astype(np.float32)
        b = ac.ConvBPDNMaskDcpl(D, Sg, 0.05, Wg, ac.ConvBPDNMaskDcpl.Options(o0), device=device)
        reg = bool(b._dev.uses_fused_rows() and b._dev.uses_fused_cols())
        out = {'workload': 'admm.cbpdn.ConvBPDNMaskDcpl %dx%d K=%d N=%d f32' % (h, w, K, N8),
               'value': rate(b, b._dev), 'unit': 'iterations/s',
               'path': 'register-resident kernels, mixed-radix lengths' if reg else 'generic chain'}
        if reg:
            del b
            os.environ['SPORCO_AMD_MD_GENERIC'] = '1'
            try:
                b = ac.ConvBPDNMaskDcpl(D, Sg, 0.05, Wg, ac.ConvBPDNMaskDcpl.Options(o0), device=device)
                out['generic_chain_value'] = rate(b, b._dev)
            finally:
                os.environ.pop('SPORCO_AMD_MD_GENERIC', None)
        return out

    def leg_pgm_mask():
        b = pc.ConvBPDNMask(D, S, 0.05, Wm, pc.ConvBPDNMask.Options(dict(o0, L=500.0)), device=device)
        return {'workload': 'pgm.cbpdn.ConvBPDNMask %dx%d K=%d N=%d f32' % (H, H, K, N8),
                'value': rate(b, b.dev), 'unit': 'iterations/s',
                'path': 'fused iteration' if b._fused_ok() else 'staged composition'}

    def leg_pgm_robust():
        n = 2 if tiny else 32
        Dc, Sc = make_problem(H, H, K, n, 0)
        b = pc.ConvBPDN(Dc, Sc, 0.05, pc.ConvBPDN.Options(dict(o0, L=500.0, Backtrack=BacktrackRobust())),
                        device=device)
        return {'workload': 'pgm.cbpdn.ConvBPDN %dx%d K=%d N=%d f32, BacktrackRobust' % (H, H, K, n),
                'value': rate(b, b.dev), 'unit': 'iterations/s',
                'path': 'fused iteration' if b._fused_ok() else 'staged composition'}

    def leg_generic(h, w, k, dt):
        def go():
            r2 = np.random.RandomState(1)
            Dg = r2.randn(8, 8, k).astype(dt)
            Sg = r2.randn(h, w, N8).astype(dt)
            b = ac.ConvBPDN(Dg, Sg, 0.05, ac.ConvBPDN.Options(o0), device=device)
            reg = bool(b._dev.uses_fused_rows() and b._dev.uses_fused_cols())
            out = {'workload': 'admm.cbpdn.ConvBPDN %dx%d K=%d N=%d %s (sizes other than 128 / 256 / 512, '
                               'or float64)' % (h, w, k, N8, np.dtype(dt).name),
                   'value': rate(b, b._dev), 'unit': 'iterations/s',
                   'path': ('register-resident kernels, mixed-radix lengths (round 6: 16 x {10 ... 30} points, '
                            'csc_rows_mr.hip)') if reg else
                           'generic chain (single-array state, fused column pass: whole tile in LDS or slabs of filters)'}
            if reg:
                # the same shape on the generic chain of this library (what served it until round 5)
                del b
                os.environ['SPORCO_AMD_UNFUSED'] = '1'
                try:
                    b = ac.ConvBPDN(Dg, Sg, 0.05, ac.ConvBPDN.Options(o0), device=device)
                finally:
                    os.environ.pop('SPORCO_AMD_UNFUSED', None)
                out['generic_chain_value'] = rate(b, b._dev)
            return out
        return go

    legs = {'maskdcpl': leg_maskdcpl, 'maskdcpl_480x320_k64_f32': leg_maskdcpl_mr, 'pgm_mask': leg_pgm_mask,
            'pgm_backtrack_robust': leg_pgm_robust}
    if tiny:
        legs['generic_48x40_k8_f32'] = leg_generic(48, 40, 8, np.float32)
        legs['generic_32x32_k8_f64'] = leg_generic(32, 32, 8, np.float64)
    else:
        # (the key names of rounds 4 / 5 are kept so that the records compare; the first three run on
        # the mixed-radix register kernels since round 6 -- `path` says which)
        legs['generic_384x384_k32_f32'] = leg_generic(384, 384, 32, np.float32)
        legs['generic_240x320_k64_f32'] = leg_generic(240, 320, 64, np.float32)
        legs['generic_480x320_k64_f32'] = leg_generic(480, 320, 64, np.float32)
        legs['generic_256x256_k32_f64'] = leg_generic(256, 256, 32, np.float64)
        legs['generic_360x360_k32_f32'] = leg_generic(360, 360, 32, np.float32)   # (no register path: 360 / 16)
    out = {}
    for name, fn in legs.items():
        try:
            out[name] = fn()
        except Exception as e:   # a failing side leg must not cost the headline line
            out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        gc.collect()
    return out


def reference_cpu_baseline(H, W, K, n_full, seconds):
    """The UNMODIFIED reference timed on this host (oracle/time_reference.py, a subprocess: this
    process never imports it).  None when oracle/_ref was not staged."""
    import subprocess
    if not os.path.isdir(os.path.join(REPO, 'oracle', '_ref', 'sporco')):
        return None
    env = dict(os.environ)
    for v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        env.setdefault(v, '1')     # (no BLAS on this path; the FFT fallback is single-threaded)
    try:
        r = subprocess.run([sys.executable, os.path.join(REPO, 'oracle', 'time_reference.py'),
                            str(H), str(W), str(K), '1,2', str(seconds)], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=40 * seconds + 120)
        j = json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e)}
    if 'error' in j:
        return j
    one, two = j['runs'][0], j['runs'][-1]
    return {'value': two['image_iterations_per_second'] / n_full, 'unit': 'iterations/s',
            'cores': 1, 'kind': 'reference',
            'sample': ('unmodified sporco.admm.cbpdn.ConvBPDN.solve() on this host (%d cores; the '
                       'reference path is single-threaded NumPy with the numpy.fft fallback -- pyFFTW '
                       'is not installed): %dx%d K=%d float32 default options, N=1: %d iterations in '
                       '%.1f s, N=2: %d iterations in %.1f s (image-iterations/s %.3f vs %.3f: linear '
                       'in N); value = N=2 image-iterations/s / %d'
                       % (j['host_cores'], H, W, K, one['iterations'], one['solve_seconds'],
                          two['iterations'], two['solve_seconds'], one['image_iterations_per_second'],
                          two['image_iterations_per_second'], n_full)),
            'detail': j}


def port_cpu_baseline(H, W, K, n_full, seconds):
    """Time the NumPy oracle (a port of the reference's arithmetic; the reference itself is
    Python and is not present on the GPU box) on ONE and on TWO images of the same workload
    (to show that the cost is linear in the number of images: they are independent and the
    arrays are far larger than cache) and scale the two-image rate by
    2/n_full.  `reference_here` quotes the unmodified reference timed on the same sample in
    the authoring container (tools/time_reference_cpu.py -> profiles/r02_reference_cpu.json),
    when that file is present."""
    from oracle import cbpdn_oracle as orc
    rates = {}
    for n, budget in ((1, seconds / 3.0), (2, 2.0 * seconds / 3.0)):
        D, S = make_problem(H, W, K, n, 0)
        r = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K), S.reshape(H, W, 1, n, 1), 0.05,
                           dtype=np.float32, maxiter=1000, rel_tol=0.0, time_budget=budget)
        rates[n] = (r['iters'], r['seconds'])
    one = rates[1][0] / rates[1][1]
    two = rates[2][0] / rates[2][1]
    # The reference's normal install threads its FFTs (pyFFTW on every core, sporco/fft.py:37)
    # while everything else stays single-threaded NumPy: the same split here, scipy.fft with
    # one worker per host core inside the oracle.
    ncores = os.cpu_count() or 1
    threaded = None
    if ncores > 1 and seconds > 0:
        orc.FFT_WORKERS = ncores
        try:
            D, S = make_problem(H, W, K, 2, 0)
            r = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K), S.reshape(H, W, 1, 2, 1), 0.05,
                               dtype=np.float32, maxiter=1000, rel_tol=0.0,
                               time_budget=seconds / 3.0)
        finally:
            orc.FFT_WORKERS = None
        threaded = {'value': (r['iters'] / r['seconds']) * 2.0 / n_full, 'unit': 'iterations/s',
                    'cores': ncores, 'kind': 'port',
                    'sample': ('the same oracle with its FFTs in scipy.fft on %d workers (the '
                               'reference threads its FFTs through pyFFTW; the elementwise NumPy '
                               'passes stay single-threaded there too): 2 images, %d iterations '
                               'in %.1f s; value = that rate x 2/%d'
                               % (ncores, r['iters'], r['seconds'], n_full))}
    out = {
        'value': two * 2.0 / n_full,
        'unit': 'iterations/s',
        'cores': 1,
        'kind': 'port',
        'sample': ('NumPy oracle (numpy.fft, single thread), %dx%d K=%d: 1 image %d iterations in '
                   '%.1f s, 2 images %d iterations in %.1f s (image-iterations/s %.3f vs %.3f: '
                   'linear in N); value = 2-image rate x 2/%d'
                   % (H, W, K, rates[1][0], rates[1][1], rates[2][0], rates[2][1], one, 2 * two,
                      n_full)),
        'why_faster_than_BASELINE_md_2': (
            'the port forms the Sherman-Morrison solve without the X-sized conj(Df)*Sf array, '
            'works in place and skips the reference\'s per-call dtype conversions and Yprev/AX '
            'copies; the unmodified reference is slower per image (see reference_here)'),
    }
    if threaded is not None:
        # the headline baseline is the faster leg; both stay in the record
        out['single_thread'] = {k: out[k] for k in ('value', 'unit', 'cores', 'kind', 'sample')}
        out['threaded_fft'] = threaded
        if threaded['value'] > out['value']:
            out.update({k: threaded[k] for k in ('value', 'cores', 'sample')})
    rpath = os.path.join(REPO, 'profiles', 'r02_reference_cpu.json')
    if os.path.exists(rpath) and (H, W, K) == (512, 512, 64):
        with open(rpath) as f:
            out['reference_here'] = json.load(f)
    return out


def cpu_baseline(H, W, K, n_full, seconds):
    """`cpu_baseline` of the line: the unmodified reference on this host when it is staged
    (`kind: "reference"`), with the NumPy port (`kind: "port"`, single-threaded and with its FFTs
    threaded over the host cores) kept beside it as `port`; the port alone otherwise."""
    ref = reference_cpu_baseline(H, W, K, n_full, seconds)
    port = port_cpu_baseline(H, W, K, n_full, seconds / 2.0)
    if ref is None or 'error' in ref:
        if ref is not None:
            port['reference_error'] = ref['error']
        return port
    ref['port'] = port
    return ref


def parity_gate(cbpdn, H, W, K, iters, device):
    """before any timing is quoted, the kernels of this workload
    (same H, W, K, dtype, options => same kernel instantiations; two of the images) against
    the float64 oracle: rel-l2 of Y and of the Rho / residual traces."""
    from oracle import cbpdn_oracle as orc
    D, S = make_problem(H, W, K, 2, 0)
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': iters,
                                                           'RelStopTol': 0.0}), device=device)
    fused = bool(b._dev.uses_fused_rows() and b._fused_ok())
    Y = b.solve().astype(np.float64)
    ref = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K), S.reshape(H, W, 1, 2, 1), 0.05,
                         dtype=np.float64, maxiter=iters, rel_tol=0.0)
    rel = float(np.linalg.norm(Y - ref['Y']) / np.linalg.norm(ref['Y']))
    its = b.getitstat()
    tr = {}
    for f in ('Rho', 'PrimalRsdl', 'DualRsdl', 'ObjFun'):
        a, r = np.asarray(getattr(its, f), float), np.asarray(ref[f], float)
        tr[f] = float(np.linalg.norm(a - r) / np.linalg.norm(r))
    del b
    return {'rel_l2_vs_oracle': rel, 'images_checked': 2, 'iters': iters,
            'oracle': 'float64 NumPy restatement (oracle/cbpdn_oracle.py)',
            'trace_rel_err': tr, 'three_launch_path': fused,
            'pass': bool(rel <= 1e-4 and max(tr.values()) <= 1e-3)}


def time_to_tol(cbpdn, H, W, K, N, device, lmbda=0.01):
    """Wall-clock and iterations to RelStopTol = 1e-3 (default options) on the
    sparse-synthesis input, N images; the same stop measured with the float64 generic chain of
    this library (pinned to the reference's traces at 1e-9 by the fixtures) as the reference
    count for this input; and the N = 2 instance whose count the unmodified reference
    produced (tests/golden/admm_tol_config2_n2_f32.npz)."""
    def run(D, S, dt):
        optd = {'MaxMainIter': 1000, 'RelStopTol': 1e-3}
        if dt is not None:
            optd['DataType'] = dt

        class Resident(cbpdn.ConvBPDN):
            def getmin(self):
                return None
        b = Resident(D, S, lmbda, cbpdn.ConvBPDN.Options(optd), device=device)
        b._dev.sync()
        t0 = time.perf_counter()
        b.solve()
        b._dev.sync()
        return b.k, time.perf_counter() - t0, bool(b._dev.uses_fused_rows() and b._fused_ok())

    D, S = make_structured_problem(H, W, K, N, 0)
    k32, t32, fused = run(D, S, None)
    k64, t64, _ = run(D, S, np.float64)
    out = {'input': 'sparse synthesis (bench.make_structured_problem), %dx%d K=%d N=%d, '
                    'lambda=%g, RelStopTol=1e-3, default options' % (H, W, K, N, lmbda),
           'iterations': k32, 'seconds': t32, 'three_launch_path': fused,
           'iterations_float64_generic_chain': k64, 'seconds_float64_generic_chain': t64}
    gpath = os.path.join(REPO, 'tests', 'golden', 'admm_tol_config2_n2_f32.npz')
    if os.path.exists(gpath) and (H, W, K) == (512, 512, 64):
        with np.load(gpath) as g:
            kref = int(g['k_final'])
        D2, S2 = make_structured_problem(H, W, K, 2, 0)
        k2, t2, _ = run(D2, S2, None)
        out['n2_case'] = {'iterations': k2, 'seconds': t2, 'reference_iterations': kref,
                          'within_one': bool(abs(k2 - kref) <= 1)}
    return out


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--filters', type=int, default=64)
    ap.add_argument('--images', type=int, default=32, help='images per GPU')
    ap.add_argument('--fastsolve', action='store_true',
                    help='FastSolve + AutoRho off as the headline run (pure iteration cost)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0,
                    help='time budget of the reference leg of cpu_baseline (the port gets half)')
    ap.add_argument('--tiny', action='store_true',
                    help='shrink the `configs` legs to simulator sizes (control-flow test only)')
    ap.add_argument('--configs', default='all',
                    help="the other BASELINE configurations to run after the headline: 'all', "
                         "'none', or a comma list of config1,config3,config4,config5")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-time-to-tol', action='store_true')
    ap.add_argument('--parity-iters', type=int, default=6)
    ap.add_argument('--steady-steps', type=int, default=100,
                    help='iterations of the same solver after the timed ones (`steady_state`)')
    ap.add_argument('--parity-only', action='store_true',
                    help='run the parity gate alone and print its JSON (what the main run spawns)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import sporco_amd
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    _lib.load()
    if sporco_amd.device_count() == 0:
        print("bench.py: no AMD GPU visible", file=sys.stderr)
        sys.exit(3)

    if os.environ.get('SPORCO_AMD_BENCH_PREALLOC_MB'):
        # (measurement knob of tools/placement_ab.py: device memory taken before the solver exists,
        # which moves every array of the run to other physical memory)
        import ctypes
        dummy = ctypes.c_void_p()
        _lib.check(_lib.lib().sporco_amd_dev_malloc(
            ctypes.c_size_t(int(os.environ['SPORCO_AMD_BENCH_PREALLOC_MB']) << 20), ctypes.byref(dummy)))

    reducer, stream, torch, dist = None, None, None, None
    if world > 1:
        import torch
        import torch.distributed as dist
        # (SPORCO_AMD_BENCH_BACKEND=gloo lets the multi-rank flow be exercised on a box with
        # fewer GPUs than ranks: ranks then share devices and reduce through host memory)
        backend = os.environ.get('SPORCO_AMD_BENCH_BACKEND', 'nccl')
        # (no GPU at all: the gloo flow on the CPU simulator build of the kernels -- the test-suite's
        # check of this file's multi-rank control flow, tests/test_dist_gloo_n.py)
        have_gpu = torch.cuda.is_available()
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        if have_gpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
        # the all-reduce of the 16 per-iteration sums: RCCL inside the library (NativeReducer:
        # enqueued by sporco_amd_csc_admm_run on the solver's stream, no Python per iteration;
        # torch.distributed only hands the communicator id round), or through torch.distributed
        # (TorchReducer: gloo runs, SPORCO_AMD_BENCH_REDUCER=torch, or when RCCL cannot be opened)
        from sporco_amd.dist import NativeReducer, TorchReducer
        which = os.environ.get('SPORCO_AMD_BENCH_REDUCER', 'native' if backend == 'nccl' else 'torch')
        if which == 'native':
            try:
                reducer = NativeReducer.from_torch(device=local_rank)
            except Exception as e:      # noqa: BLE001 -- any failure here must not cost the run
                print('bench.py: native RCCL reducer unavailable (%s); using torch.distributed' % e,
                      file=sys.stderr)
                reducer = None
        if reducer is None:
            reducer = TorchReducer()
        stream = reducer.stream_handle()

    H = W = args.size
    K, N = args.filters, args.images

    # parity gate first, rank 0, on the kernels this workload runs -- in a process
    # of its own, so that the timed solver meets the device memory as a fresh process does (round 4
    # did this because the allocation history decided between three speeds of the dominant kernel;
    # since round 5 the library places its concurrently written arrays by measurement --
    # DESIGN.md 4.7, `roofline.placement` below -- and the separation only keeps the runs alike)
    parity = None
    if rank == 0 and not args.no_parity:
        if args.parity_only:
            print(json.dumps(parity_gate(cbpdn, H, W, K, args.parity_iters, local_rank)))
            return
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--parity-only', '--size', str(H),
                                '--filters', str(K), '--parity-iters', str(args.parity_iters)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                               env=dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK=str(local_rank)))
            parity = json.loads(r.stdout.decode().strip().splitlines()[-1])
        except Exception as e:      # noqa: BLE001 -- fall back to the in-process gate
            print('bench.py: parity gate subprocess failed (%s); running it in-process' % e, file=sys.stderr)
            parity = parity_gate(cbpdn, H, W, K, args.parity_iters, local_rank)

    D, S = make_problem(H, W, K, N, rank)

    class ResidentConvBPDN(cbpdn.ConvBPDN):
        """solve() normally returns the coefficient array, i.e. ends with a
        device-to-host copy of Y (2.1 GB here).  The metric is defined with the
        data resident in HBM, so the timed solver leaves Y on the device; the
        PCIe-inclusive figure is reported separately (`result_download_ms`)."""

        def getmin(self):
            return None

    def sync_all(b):
        b._dev.sync()
        if world > 1:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dist.barrier()

    rank_stats = {}

    def timed_run(fast):
        optd = {'MaxMainIter': max(args.warmup, 1), 'RelStopTol': 0.0}
        if fast:
            optd.update({'FastSolve': True, 'AutoRho': {'Enabled': False}})
        b = ResidentConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd), device=local_rank,
                             stream=stream, reducer=reducer)
        if args.warmup > 0:
            b.solve()
        # timed region: exactly `steps` iterations, no instrumentation (and no cyclic-garbage
        # collection: see timed_solve)
        b.opt['MaxMainIter'] = args.steps
        gc.collect()
        gc.disable()
        try:
            sync_all(b)
            t0 = time.perf_counter()
            b.solve()
            sync_all(b)
            elapsed = time.perf_counter() - t0
        finally:
            gc.enable()
        if world > 1:
            own = elapsed
            dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.cpu()[0])
            # per-rank spread and the cost of the per-iteration all-reduce, so that a scaling
            # shortfall can be attributed (slow rank / collective / neither)
            t = torch.tensor([-own], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            hc = reducer.hook_cost_ms()
            h = torch.tensor([hc[0] if hc else 0.0, hc[1] if hc else 0.0], dtype=torch.float64,
                             device=dev)
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            if not rank_stats:
                rank_stats.update({
                    'ms_per_step_slowest_rank': 1e3 * elapsed / args.steps,
                    'ms_per_step_fastest_rank': 1e3 * -float(t.cpu()[0]) / args.steps,
                    'allreduce_ms_mean_worst_rank': float(h.cpu()[0]),
                    'allreduce_ms_max': float(h.cpu()[1]),
                    'allreduce_calls_timed': hc[2] if hc else 0,
                    'backend': dist.get_backend(),
                    'reducer': type(reducer).__name__})
        return b, elapsed

    b, elapsed = timed_run(args.fastsolve)
    # the same solver object carried on for 100 more iterations: with default options rho moves
    # in most of the first 20-30 iterations (which invalidates the speculatively emitted row
    # spectra); what a user sees over the hundreds of iterations to tolerance is the rate after
    # it has settled
    steady_steps = max(args.steady_steps, 1)
    b.opt['MaxMainIter'] = steady_steps
    sync_all(b)
    t0s = time.perf_counter()
    b.solve()
    sync_all(b)
    elapsed_steady = time.perf_counter() - t0s
    if world > 1:
        ts = torch.tensor([elapsed_steady], dtype=torch.float64,
                          device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        elapsed_steady = float(ts.cpu()[0])
    # per-kernel durations: HIP events recorded by the library on its own stream
    # around every launch, over a second run of the same iterations
    # (this pass runs the host-driven loop, which launches exactly the kernels that execute:
    # the device-driven loop of the timed region enqueues both epilogue variants and a forward
    # row pass every iteration and lets the device skip the ones not needed, so its event
    # timings would average idle launches in)
    prof_steps = min(args.steps, 10)
    b.opt['MaxMainIter'] = prof_steps
    b.profile(True)
    os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
    try:
        b.solve()
    finally:
        os.environ.pop('SPORCO_AMD_HOST_LOOP', None)
    sync_all(b)
    prof = b.profile_read()
    b.profile(False)
    # the emitting epilogue in each of its two ping-pong directions (V buffer A -> B, B -> A): one
    # iteration per solve() call, HIP events of that launch alone.  Equal times = both iterate
    # buffers lie clear of the spectrum buffer; one direction ~8 % slower = that buffer shares its
    # region whatever the placement report says.
    per_direction = None
    if world == 1:
        dirs = [[], []]
        b.opt['MaxMainIter'] = 1
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
        try:
            for i in range(8):
                b.profile(True)
                b.solve()
                b._dev.sync()
                p1 = b.profile_read()
                b.profile(False)
                v = p1.get('rows_inv_post_v_emit')
                if v and v[1] == 1:
                    dirs[i & 1].append(v[0])
        finally:
            os.environ.pop('SPORCO_AMD_HOST_LOOP', None)
        if dirs[0] and dirs[1]:
            per_direction = [round(float(np.median(d)), 4) for d in dirs]
    # where the library put the arrays the dominant kernel writes at the same time
    # (sporco_amd_csc_placement_report; profiles/r05_placement_notes.md)
    placement = b._dev.placement_report()
    t1 = time.perf_counter()
    y_host = cbpdn.ConvBPDN.getmin(b)          # what solve() would hand back to the caller
    download_ms = 1e3 * (time.perf_counter() - t1)
    del y_host
    # the other option set as a secondary figure (both are reported); its solver
    # is created while the first one still holds its arrays, i.e. in memory nobody has freed
    b2, elapsed2 = timed_run(not args.fastsolve)
    del b2
    del b

    config3_sharded = None
    if world > 1:
        dev_t = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'

        def allmax(x):
            t = torch.tensor([float(x)], dtype=torch.float64, device=dev_t)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.cpu()[0])

        # the same local problem WITHOUT the reducer on every rank at once: what one GPU does alone
        # -- the N = 1 figure of this very invocation.  sharded / unsharded is the cost of the
        # collective and of waiting for the slowest rank; the driver's N = 1 run must agree with
        # `unsharded` (same code path: a one-rank invocation has no reducer either).
        bu = ResidentConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(
            {'MaxMainIter': max(args.warmup, 1), 'RelStopTol': 0.0}), device=local_rank, stream=stream)
        bu.solve()
        bu.opt['MaxMainIter'] = args.steps
        sync_all(bu)
        t0u = time.perf_counter()
        bu.solve()
        bu._dev.sync()
        own_u = time.perf_counter() - t0u
        del bu
        slow_u, fast_u = allmax(own_u), -allmax(-own_u)
        rank_stats.update({'unsharded_ms_per_step_slowest_rank': 1e3 * slow_u / args.steps,
                           'unsharded_ms_per_step_fastest_rank': 1e3 * fast_u / args.steps,
                           'sharded_over_unsharded_time': elapsed / slow_u})
        if args.configs != 'none' and (args.configs == 'all' or 'config3' in args.configs.split(',')):
            gc.collect()
            try:
                config3_sharded = run_config3_sharded(local_rank, rank, world, reducer, stream, sync_all,
                                                      allmax, tiny=args.tiny)
            except Exception as e:      # noqa: BLE001 -- every rank fails alike (same shapes) or none
                config3_sharded = {'error': '%s: %s' % (type(e).__name__, e)}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    its_per_s = args.steps / elapsed
    moved, alg = byte_model(H, W, 1, N, K)
    table = kernel_roofline(prof, moved, alg)
    dom = dominant(table)
    dom_ms = table[dom]['avg_ms']
    # what the committed profiles say about the same kernel of the same command (rocprofv3
    # --kernel-trace --stats average of the working dispatches; PMC FETCH_SIZE / WRITE_SIZE passes):
    # profiles/hbm_traffic_bytes.json, produced by tools/final_prof.sh -- every figure of
    # `from_profiles` is recomputable from that one file
    traffic, from_profiles = None, None
    tpath = os.path.join(REPO, 'profiles', 'hbm_traffic_bytes.json')
    if os.path.exists(tpath) and (H, W, K, N) == (512, 512, 64, 32):
        with open(tpath) as f:
            tj = json.load(f)
        traffic = tj.get(dom)
        rp_ms = tj.get('_rocprof_avg_ms', {}).get(dom)
        if traffic and rp_ms:
            from_profiles = {'file': 'profiles/hbm_traffic_bytes.json', 'kernel': dom,
                             'traffic_bytes_per_launch': traffic, 'rocprof_avg_kernel_ms': rp_ms,
                             'achieved': traffic / rp_ms / 1e6,
                             'frac': traffic / rp_ms / 1e6 / HBM_PEAK_GBPS,
                             'note': 'PMC traffic (2*FETCH_SIZE + WRITE_SIZE)*1024 per working '
                                     'dispatch / rocprofv3 --kernel-trace average of the same '
                                     'kernel in `python bench.py`; not re-measured in this run '
                                     '(boxes of the pool differ by up to 8 %)'}
    E = H * W * N * K
    iter_alg_bytes = 40 * E                    # 10 float32 passes
    names = {False: 'default options (AutoRho, stats every iteration)',
             True: 'FastSolve, AutoRho off'}
    line = {
        'metric': 'ConvBPDN ADMM iterations/s',
        'value': its_per_s * world,
        'unit': 'iterations/s (%dx%d, K=%d, N=%d per GPU; summed over GPUs)' % (H, W, K, N),
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': 'admm.cbpdn.ConvBPDN %dx%d greyscale, K=%d 8x8 filters, '
                               'N=%d images per GPU, lambda=0.05, %s'
                               % (H, W, K, N, names[bool(args.fastsolve)]),
                   'global_images': N * world, 'parallelism': 'image-shard x%d' % world},
        'steady_state': {'steps': steady_steps, 'value': steady_steps / elapsed_steady * world,
                         'ms_per_step': 1e3 * elapsed_steady / steady_steps,
                         'note': 'the same solver continued for %d more iterations after the timed '
                                 '%d (rho has settled: the speculatively emitted row spectra hold)'
                                 % (steady_steps, args.steps)},
        'roofline': {'bound': 'hbm', 'kernel': dom,
                     # bytes the kernel MOVES (inputs + outputs once; = PMC traffic) / its average
                     # launch duration by HIP events on the library's stream in this run
                     'achieved': moved[dom] / dom_ms / 1e6, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                     'frac': moved[dom] / dom_ms / 1e6 / HBM_PEAK_GBPS,
                     'traffic': traffic, 'avg_kernel_ms': dom_ms,
                     'bytes_per_launch': moved[dom],
                     'bytes_basis': 'moved by construction (byte_model); the measured PMC traffic '
                                    'of profiles/hbm_traffic_bytes.json agrees to 0.1 %',
                     # SURVEY.md 8(d)'s stage accounting of the same kernel (for the kernels of
                     # the single-array state it credits bytes that are not moved: not a
                     # distance to the HBM peak)
                     'algorithmic_bytes_per_launch': alg[dom],
                     'algorithmic_achieved': alg[dom] / dom_ms / 1e6,
                     'stage_model_ratio': alg[dom] / dom_ms / 1e6 / HBM_PEAK_GBPS,
                     'from_profiles': from_profiles,
                     # (scalars first: consumers that flatten this object keep them)
                     'per_direction_ms': per_direction,
                     'placement_clear_all': bool(all(d.get('clear') for d in placement)) if placement else None,
                     'placement_compact': ' '.join('%s:%s/%d@%.2f' % (d.get('role'), 'clear' if d.get('clear') else 'SHARED',
                                                                      d.get('candidates', 0), d.get('chosen_ratio', 0.0))
                                                   for d in placement) if placement else '',
                     'placement': placement},
        'iteration_roofline': dict(
            iteration_summary(table, prof_steps, ms_per_step, iter_alg_bytes),
            steady_state_stage_model_ratio=iter_alg_bytes * (steady_steps / elapsed_steady) / 1e9
            / HBM_PEAK_GBPS),
        'other_options': {'options': names[not args.fastsolve],
                          'value': args.steps / elapsed2 * world,
                          'ms_per_step': 1e3 * elapsed2 / args.steps},
        'ranks': rank_stats or None,
        'parity': parity,
        'loop': ('device-driven (sporco_amd_csc_admm_run): residuals, rho schedule and stopping '
                 'test on the device, host enqueues only') if not os.environ.get('SPORCO_AMD_HOST_LOOP')
                else 'host-driven (one sporco_amd_csc_admm_iter call per iteration)',
        'result_download_ms': download_ms,
        'kernels_ms_per_iter': {k: round(v[0] / prof_steps, 4) for k, v in prof.items() if v[1] > 0},
        'kernel_roofline': table,
    }
    if config3_sharded is not None:
        line['configs'] = {'config3_sharded': config3_sharded}
    if world == 1 and args.configs != 'none':
        line['configs'] = run_other_configs(local_rank, args.configs, tiny=args.tiny)
    if world == 1 and args.configs == 'all':
        line['next_rows'] = run_next_rows(local_rank, tiny=args.tiny)
    if world == 1 and not args.no_time_to_tol:
        line['time_to_tol'] = time_to_tol(cbpdn, H, W, K, N, local_rank)
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(H, W, K, N, args.cpu_seconds)
    bad = check_fracs(line)
    if os.environ.get('SPORCO_AMD_BENCH_STRICT'):
        assert not bad, 'fraction of a peak above 1: %s' % bad
    line['frac_check'] = {'violations': bad, 'rule': 'no `frac` of this line exceeds 1'}
    assert not bad or not os.environ.get('SPORCO_AMD_BENCH_STRICT')
    if bad:      # (never print one: blank the offenders, keep the line)
        def blank(obj):
            if isinstance(obj, dict):
                for k in list(obj):
                    if k == 'frac' and isinstance(obj[k], (int, float)) and obj[k] > 1.0:
                        obj[k] = None
                    else:
                        blank(obj[k])
            elif isinstance(obj, list):
                for v in obj:
                    blank(v)
        blank(line)
    # the numbers that matter once more, compact, as the LAST key: a consumer that keeps only the
    # tail of this (long) line still has them
    cfg = line.get('configs') or {}
    line['summary'] = {
        'value': line['value'], 'ms_per_step': ms_per_step, 'steady_state': line['steady_state']['value'],
        'roofline_kernel': dom, 'roofline_frac': line['roofline']['frac'], 'avg_kernel_ms': dom_ms,
        'placement_clear_all': line['roofline']['placement_clear_all'],
        'placement': line['roofline']['placement_compact'], 'per_direction_ms': per_direction,
        'configs': {k: (round(v['value'], 2) if isinstance(v, dict) and 'value' in v else None)
                    for k, v in cfg.items()},
        'config_roofline_frac': {k: v['roofline']['frac'] for k, v in cfg.items()
                                 if isinstance(v, dict) and isinstance(v.get('roofline'), dict)},
        'next_rows': {k: (round(v['value'], 1) if isinstance(v, dict) and 'value' in v else None)
                      for k, v in (line.get('next_rows') or {}).items()},
        'parity_pass': (parity or {}).get('pass') if isinstance(parity, dict) else None,
        'cpu_baseline': (line.get('cpu_baseline') or {}).get('value'),
        'ranks': ({k: rank_stats[k] for k in ('ms_per_step_slowest_rank', 'ms_per_step_fastest_rank',
                                              'sharded_over_unsharded_time', 'reducer') if k in rank_stats}
                  if rank_stats else None),
        'frac_violations': bad}
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
