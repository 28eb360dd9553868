"""Worker of tests/test_dist_gloo_joint.py: one rank of an image-sharded ConvBPDNJoint solve --
the class BASELINE configs[2] names for the 8-GPU workload (sporco/admm/cbpdn.py:636-807; the
l2,1 term couples the CHANNELS of a pixel, never the images, so images shard as for ConvBPDN).
CPU: gloo process group + the fiber-simulator build of the kernels."""

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

F = ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')


def traces(its, tag):
    return {tag + f: np.asarray(getattr(its, f), dtype=float) for f in F}


def main():
    out_path = sys.argv[1]
    import torch.distributed as dist
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    import sporco_amd
    from conftest import HOSTSIM_LIB, load_golden
    sporco_amd.load_library(HOSTSIM_LIB)
    from sporco_amd.admm import cbpdn
    from sporco_amd.dist import TorchReducer, shard_bounds, shard_images
    out = {}

    # 1. the reference's own two-image run (tests/golden/admm_joint_f64, written by the unmodified
    #    reference), one image per rank -- only at world size 2
    if world == 2:
        g = load_golden('admm_joint_f64')
        opt = cbpdn.ConvBPDNJoint.Options({'MaxMainIter': int(g['k_final'])})
        b = cbpdn.ConvBPDNJoint(g['D'], shard_images(g['S'], rank, world), float(g['lmbda']), float(g['mu']),
                                opt, reducer=TorchReducer())
        out['ref_Y'] = b.solve()
        out['ref_k'] = b.k
        out.update(traces(b.getitstat(), 'ref_'))

    # 2. uneven shards (rank 0 holds two images, every other rank one), float64 generic chain,
    #    default AutoRho: every rank must move rho alike
    n_img = world + 1
    lo, hi = shard_bounds(n_img, rank, world)
    out['lo'], out['hi'] = lo, hi
    rng = np.random.RandomState(2718)
    D = rng.randn(5, 5, 6)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(16, 12, 3, n_img)
    o = {'MaxMainIter': 14, 'RelStopTol': 1e-3}
    b = cbpdn.ConvBPDNJoint(D, shard_images(S, rank, world), 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(o),
                            reducer=TorchReducer())
    out['g_Y'] = b.solve()
    out['g_k'] = b.k
    out.update(traces(b.getitstat(), 'g_'))
    if rank == 0:
        b1 = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(o))
        out['g1_Y'] = b1.solve()
        out['g1_k'] = b1.k
        out.update(traces(b1.getitstat(), 'g1_'))

    # 3. the register-resident joint kernels (float32, 128 x 128 x 3, K = 32: lane = (channel, filter
    #    pair)), device-driven loop with the all-reduce between the local sums and the control
    #    kernel, unequal host lag; then the host-driven loop of the same sharded problem
    Df = rng.randn(4, 4, 32).astype(np.float32)
    Df /= np.sqrt(np.sum(Df ** 2, axis=(0, 1), keepdims=True))
    Sf = rng.randn(128, 128, 3, n_img).astype(np.float32)
    of = {'MaxMainIter': 4, 'RelStopTol': 0.0}
    os.environ['SPORCO_AMD_RUN_LAG'] = '2' if rank % 2 else '0'
    red = TorchReducer()
    bf = cbpdn.ConvBPDNJoint(Df, shard_images(Sf, rank, world), 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(of),
                             reducer=red)
    assert bf._device_loop_ok() and bf._dev.uses_fused_rows()
    assert bf._reducer.device_sum_hook(bf._dev) is not None
    out['d_Y'] = bf.solve()
    os.environ.pop('SPORCO_AMD_RUN_LAG')
    out['d_after'] = red.sum([float(rank + 1)])[0]
    out['d_k'] = bf.k
    out.update(traces(bf.getitstat(), 'd_'))
    os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
    bh = cbpdn.ConvBPDNJoint(Df, shard_images(Sf, rank, world), 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(of),
                             reducer=TorchReducer())
    out['h_Y'] = bh.solve()
    out.update(traces(bh.getitstat(), 'h_'))
    if rank == 0:
        b1 = cbpdn.ConvBPDNJoint(Df, Sf, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(of))
        assert b1._dev.uses_fused_rows()
        out['d1_Y'] = b1.solve()
        out.update(traces(b1.getitstat(), 'd1_'))
    os.environ.pop('SPORCO_AMD_HOST_LOOP')
    np.savez(out_path + '.%d.npz' % rank, **out)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
