"""Worker of tests/test_dist_nccl.py::test_two_ranks_share_one_gpu: one rank of a 2-process
image-sharded solve in which BOTH ranks run the HIP library on GPU 0 and reduce through gloo
(RCCL does not accept two ranks on one device).  Correctness only: it exercises the
device-driven loop with a real all-reduce hook between the sums and the control kernel on
real hardware -- host-staged (sporco_amd/dist.py device_sum_hook) -- including an early
tolerance stop with unequal host lags."""

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    out_path = sys.argv[1]
    import torch.distributed as dist
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    import sporco_amd
    from sporco_amd import _lib
    _lib.load()
    assert 'hostsim' not in str(_lib.library_path()) and sporco_amd.device_count() > 0
    from sporco_amd.admm import cbpdn
    from sporco_amd.dist import TorchReducer, shard_images
    from test_fused_xstep import problem
    D, S = problem(256, 256, 8, 2, seed=4)
    res = {}
    for name, optd, lag in (('fixed', {'MaxMainIter': 8, 'RelStopTol': 0.0}, 0),
                            ('early', {'MaxMainIter': 60, 'RelStopTol': 2e-2}, 3 if rank else 0)):
        red = TorchReducer()
        assert not red.on_gpu
        os.environ['SPORCO_AMD_RUN_LAG'] = str(lag)
        b = cbpdn.ConvBPDN(D, shard_images(S, rank, world, axis=-1), 0.05,
                           cbpdn.ConvBPDN.Options(optd), device=0, reducer=red)
        assert b._device_loop_ok() and b._dev.uses_fused_rows()
        hook = red.device_sum_hook(b._dev)
        assert hook is not None
        Y = b.solve()
        os.environ.pop('SPORCO_AMD_RUN_LAG')
        after = red.sum([float(rank + 1)])[0]
        its = b.getitstat()
        res[name + '_Y'] = Y
        res[name + '_k'] = b.k
        res[name + '_after'] = after
        for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            res[name + '_' + f] = np.asarray(getattr(its, f), dtype=float)
        if rank == 0:       # the single-process run of both images
            b1 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd), device=0)
            res[name + '_Y1'] = b1.solve()
            res[name + '_k1'] = b1.k
            i1 = b1.getitstat()
            for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
                res[name + '_' + f + '1'] = np.asarray(getattr(i1, f), dtype=float)
    np.savez(out_path + '.%d.npz' % rank, **res)
    dist.barrier()
    dist.destroy_process_group()
    print('GLOO_GPU_WORKER_OK rank %d' % rank)


if __name__ == '__main__':
    main()
