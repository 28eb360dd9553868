#!/usr/bin/env python3
"""bench.py -- ConvBPDN ADMM iterations/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
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
"""

import argparse
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


def kernel_bytes(H, W, P, itemsize):
    """Compulsory HBM bytes (inputs + outputs once) of each kernel of one ADMM
    iteration; P = C*N*K.  See DESIGN.md section 5."""
    E = H * W * P * itemsize                 # one pass over a real X-sized array
    EF = H * (W // 2 + 1) * P * 2 * itemsize  # one pass over a half-spectrum array
    return {
        'fft_r2c_rows': 2 * E + EF,          # read Y, U; write row spectra
        'fft_c2c_cols_fwd': 2 * EF,
        'sm_solve': 2 * EF,
        'fused_cols_sm': 2 * EF,             # read row spectra, write solved column-IFFT (in place)
        'fft_c2c_cols_inv': 2 * EF,
        'fft_c2r_rows': EF + E,              # read spectra; write X
        'admm_post': 5 * E,                  # read X, Y, U; write Y, U
        'rows_fwd': 2 * E + EF,              # read Y, U; write tile-major row spectra
        'rows_inv_post': EF + 4 * E,         # read spectra, Y, U; write Y, U (X stays in registers)
        'rows_inv_post_emit': 2 * EF + 4 * E,  # ... and write the next iteration's row spectra
    }


def cpu_baseline(H, W, K, n_full, seconds):
    """Time the NumPy oracle (a port of the reference's arithmetic; the
    reference itself is not present on the GPU box) on ONE image of the same
    workload and scale by 1/n_full (images are independent and the arrays are
    far larger than cache)."""
    from oracle import cbpdn_oracle as orc
    D, S = make_problem(H, W, K, 1, 0)
    r = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K), S.reshape(H, W, 1, 1, 1), 0.05,
                       dtype=np.float32, maxiter=1000, rel_tol=0.0, time_budget=seconds)
    its_per_s_one = r['iters'] / r['seconds']
    return {
        'value': its_per_s_one / n_full,
        'unit': 'iterations/s',
        'cores': 1,
        'kind': 'port',
        'sample': ('NumPy oracle (numpy.fft, single thread), %dx%d K=%d on 1 of %d images, '
                   '%d iterations in %.1f s, rate divided by %d'
                   % (H, W, K, n_full, r['iters'], r['seconds'], n_full)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--filters', type=int, default=64)
    ap.add_argument('--images', type=int, default=32, help='images per GPU')
    ap.add_argument('--fastsolve', action='store_true',
                    help='FastSolve + AutoRho off (pure iteration cost, no stats)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch through "
                  "torch.distributed.run for N > 1" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import sporco_amd
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    _lib.load()
    if sporco_amd.device_count() == 0:
        print("bench.py: no AMD GPU visible", file=sys.stderr)
        sys.exit(3)

    reducer, stream, torch, dist = None, None, None, None
    if world > 1:
        import torch
        import torch.distributed as dist
        # (SPORCO_AMD_BENCH_BACKEND=gloo lets the multi-rank flow be exercised on a box with
        # fewer GPUs than ranks: ranks then share devices and reduce through host memory)
        backend = os.environ.get('SPORCO_AMD_BENCH_BACKEND', 'nccl')
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
        from sporco_amd.dist import TorchReducer
        reducer = TorchReducer()
        stream = reducer.stream_handle()

    H = W = args.size
    K, N = args.filters, args.images
    D, S = make_problem(H, W, K, N, rank)
    optd = {'MaxMainIter': max(args.warmup, 1), 'RelStopTol': 0.0}
    if args.fastsolve:
        optd.update({'FastSolve': True, 'AutoRho': {'Enabled': False}})

    class ResidentConvBPDN(cbpdn.ConvBPDN):
        """solve() normally returns the coefficient array, i.e. ends with a
        device-to-host copy of Y (2.1 GB here).  The metric is defined with the
        data resident in HBM, so the timed solver leaves Y on the device; the
        PCIe-inclusive figure is reported separately (`result_download_ms`)."""

        def getmin(self):
            return None

    b = ResidentConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd), device=local_rank,
                         stream=stream, reducer=reducer)

    def sync_all():
        b._dev.sync()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()

    if args.warmup > 0:
        b.solve()
    # timed region: exactly `steps` iterations, no instrumentation
    b.opt['MaxMainIter'] = args.steps
    sync_all()
    t0 = time.perf_counter()
    b.solve()
    sync_all()
    elapsed = time.perf_counter() - t0
    # per-kernel durations: HIP events recorded by the library on its own stream
    # around every launch, over a second run of the same iterations
    prof_steps = min(args.steps, 10)
    b.opt['MaxMainIter'] = prof_steps
    b.profile(True)
    b.solve()
    sync_all()
    prof = b.profile_read()
    b.profile(False)
    t1 = time.perf_counter()
    y_host = cbpdn.ConvBPDN.getmin(b)          # what solve() would hand back to the caller
    download_ms = 1e3 * (time.perf_counter() - t1)
    del y_host
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.cpu()[0])

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    P = N * K
    itemsize = 4
    ms_per_step = 1e3 * elapsed / args.steps
    its_per_s = args.steps / elapsed
    kb = kernel_bytes(H, W, P, itemsize)
    timed = {k: v for k, v in prof.items() if v[1] > 0}
    dom = max((k for k in timed if k in kb), key=lambda k: timed[k][0])
    dom_ms = timed[dom][0] / timed[dom][1]
    achieved = kb[dom] / (dom_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'hbm_traffic_bytes.json')
    if os.path.exists(tpath) and (H, W, K, N) == (512, 512, 64, 32):
        # measured for exactly this workload (rocprofv3 PMC passes, see the file)
        with open(tpath) as f:
            traffic = json.load(f).get(dom)
    E = H * W * P
    iter_alg_bytes = 40 * E                    # 10 float32 passes
    line = {
        'metric': 'ConvBPDN ADMM iterations/s',
        'value': its_per_s * world,
        'unit': 'iterations/s (512x512, K=64, N=32 per GPU; summed over GPUs)',
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
                               % (H, W, K, N, 'FastSolve, AutoRho off' if args.fastsolve else
                                  'default options (AutoRho, stats every iteration)'),
                   'global_images': N * world, 'parallelism': 'image-shard x%d' % world},
        'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': achieved,
                     'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                     'traffic': traffic, 'avg_kernel_ms': dom_ms,
                     'algorithmic_bytes_per_launch': kb[dom]},
        'iteration_roofline': {'algorithmic_bytes_per_iter': iter_alg_bytes,
                               'achieved': iter_alg_bytes * its_per_s / 1e9,
                               'unit': 'GB/s',
                               'frac': iter_alg_bytes * its_per_s / 1e9 / HBM_PEAK_GBPS},
        'result_download_ms': download_ms,
        'kernels_ms_per_iter': {k: round(v[0] / prof_steps, 4) for k, v in timed.items()},
    }
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(H, W, K, N, args.cpu_seconds)
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
