"""Worker of tests/test_dist_nccl.py: a one-rank RCCL ('nccl') process group on the GPU.

Exercises what the multi-GPU bench relies on and a single-GPU run never touches:
torch and libsporco_amd.so in one process (one shared HIP runtime), the reducer's
device buffer written by the solver's stream and reduced by RCCL, the shared stream."""

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    import sporco_amd
    from sporco_amd import _lib
    _lib.load()                          # before torch on purpose: the harder order
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    from sporco_amd.admm import cbpdn
    from sporco_amd.dist import TorchReducer
    from test_fused_xstep import problem
    from conftest import rel_l2
    D, S = problem(256, 256, 8, 2, seed=4)
    optd = {'MaxMainIter': 8, 'RelStopTol': 0.0}
    red = TorchReducer()
    assert red.on_gpu
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd), stream=red.stream_handle(),
                       reducer=red)
    Y = b.solve()
    b0 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    Y0 = b0.solve()
    assert np.array_equal(Y, Y0), rel_l2(Y, Y0)
    assert np.array_equal(np.asarray(b.getitstat().Rho), np.asarray(b0.getitstat().Rho))
    # ... and with the solver on a stream of its own (not torch's current one): the hook issues
    # the collective on that stream through torch.cuda.ExternalStream, so the ordering between
    # the kernels that produce the sums, RCCL and the control kernel does not depend on which
    # stream the handle was created with
    b2 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd), reducer=red)
    assert b2._dev.stream_handle() != red.stream_handle()
    assert red.device_sum_hook(b2._dev) is not None
    Y2 = b2.solve()
    assert np.array_equal(Y2, Y0), rel_l2(Y2, Y0)
    assert np.array_equal(np.asarray(b2.getitstat().Rho), np.asarray(b0.getitstat().Rho))
    # dictionary learning with the reducer: the D-step gradient is all-reduced in place in the
    # library's device memory (one rank: the sum is the identity, the plumbing is what runs)
    from sporco_amd.dictlrn import cbpdndl
    rng = np.random.RandomState(9)
    D0, Sd = rng.randn(6, 6, 8), rng.randn(256, 256, 2).astype(np.float32)
    outs = []
    for r in (red, None):
        opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 4, 'AccurateDFid': True},
                                                xmethod='admm', dmethod='pgm')
        kw = {} if r is None else {'reducer': r, 'stream': r.stream_handle()}
        d = cbpdndl.ConvBPDNDictLearn(D0.astype(np.float32), Sd, 0.1, opt, xmethod='admm',
                                      dmethod='pgm', **kw)
        outs.append((d.solve(), np.asarray(d.getitstat().ObjFun, dtype=float)))
    assert np.array_equal(outs[0][0], outs[1][0]), rel_l2(outs[0][0], outs[1][0])
    assert rel_l2(outs[0][1], outs[1][1]) < 1e-12
    dist.destroy_process_group()
    print('NCCL_WORKER_OK transport=%s' % getattr(red, 'array_transport', None))


if __name__ == '__main__':
    main()
