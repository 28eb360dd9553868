"""Worker of tests/test_dist_gloo_n.py: one rank of a WORLD-process image-sharded solve with
UNEVEN shards (CPU: gloo process group + the fiber-simulator build of the kernels).  Rank 0 also
runs every problem unsharded; the test compares."""

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def traces(its, fields):
    return {f: np.asarray(getattr(its, f), dtype=float) for f in fields}


def main():
    out_path = sys.argv[1]
    import torch.distributed as dist
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    import sporco_amd
    from conftest import HOSTSIM_LIB
    sporco_amd.load_library(HOSTSIM_LIB)
    from sporco_amd.admm import cbpdn
    from sporco_amd.dictlrn import cbpdndl
    from sporco_amd.dist import TorchReducer, shard_bounds, shard_images
    n_img = world + 1                        # rank 0 holds two images, every other rank one
    lo, hi = shard_bounds(n_img, rank, world)
    out = {'lo': lo, 'hi': hi}
    rng = np.random.RandomState(4711)

    # 1. float64, generic chain, default options (AutoRho moves rho: every rank must move it alike)
    D = rng.randn(4, 4, 5)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(16, 12, n_img)
    F = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')
    opt = {'MaxMainIter': 12, 'RelStopTol': 1e-3}
    b = cbpdn.ConvBPDN(D, shard_images(S, rank, world), 0.05, cbpdn.ConvBPDN.Options(opt),
                       reducer=TorchReducer())
    out['g_Y'] = b.solve()
    out['g_k'] = b.k
    out.update({'g_' + k: v for k, v in traces(b.getitstat(), F).items()})
    if rank == 0:
        b1 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(opt))
        out['g1_Y'] = b1.solve()
        out['g1_k'] = b1.k
        out.update({'g1_' + k: v for k, v in traces(b1.getitstat(), F).items()})

    # 2. float32, register-resident kernels, device-driven loop, stopping early, the odd ranks' hosts
    #    three records late: every rank issues the same number of collectives
    Df = rng.randn(4, 4, 4).astype(np.float32)
    Df /= np.sqrt(np.sum(Df ** 2, axis=(0, 1), keepdims=True))
    Sf = rng.randn(128, 128, n_img).astype(np.float32)
    opts = {'MaxMainIter': 24, 'RelStopTol': 8e-2}
    os.environ['SPORCO_AMD_RUN_LAG'] = '3' if rank % 2 else '0'
    red = TorchReducer()
    be = cbpdn.ConvBPDN(Df, shard_images(Sf, rank, world), 0.05, cbpdn.ConvBPDN.Options(opts), reducer=red)
    assert be._device_loop_ok() and be._reducer.device_sum_hook(be._dev) is not None
    out['d_Y'] = be.solve()
    os.environ.pop('SPORCO_AMD_RUN_LAG')
    out['d_after'] = red.sum([float(rank + 1)])[0]      # world (world + 1) / 2 only if aligned
    out['d_k'] = be.k
    FD = ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho')
    out.update({'d_' + k: v for k, v in traces(be.getitstat(), FD).items()})
    if rank == 0:
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
        b1 = cbpdn.ConvBPDN(Df, Sf, 0.05, cbpdn.ConvBPDN.Options(opts))
        out['d1_Y'] = b1.solve()
        out['d1_k'] = b1.k
        out.update({'d1_' + k: v for k, v in traces(b1.getitstat(), FD).items()})
        os.environ.pop('SPORCO_AMD_HOST_LOOP')

    # 3. dictionary learning: ADMM X-step + PGM D-step (gradient all-reduced, default L from the
    #    GLOBAL image count), then the consensus D-step (weighted average over unequal shards)
    D0 = rng.randn(4, 4, 6)
    Sd = rng.randn(16, 16, n_img)
    for dm in ('pgm', 'cns'):
        o = {'MaxMainIter': 2, 'AccurateDFid': True}
        runs = [(shard_images(Sd, rank, world), {'reducer': TorchReducer()})]
        if rank == 0:
            runs.append((Sd, {}))
        for i, (Si, kw) in enumerate(runs):
            opt = cbpdndl.ConvBPDNDictLearn.Options(o, xmethod='admm', dmethod=dm)
            d = cbpdndl.ConvBPDNDictLearn(D0, Si, 0.1, opt, xmethod='admm', dmethod=dm, **kw)
            tag = 'dl%s%s_' % (dm, '1' if i else '')
            out[tag + 'D'] = d.solve()
            out[tag + 'X'] = d.getcoef()
            its = d.getitstat()
            out.update({tag + f: np.asarray(getattr(its, f), dtype=float) for f in its._fields
                        if f not in ('Iter', 'Time')})
    np.savez(out_path + '.%d.npz' % rank, **out)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
