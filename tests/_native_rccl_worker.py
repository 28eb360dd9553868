"""Worker of tests/test_dist_native.py: the library's own RCCL communicator (sporco_amd_comm_*,
csc_comm.hip) with ONE rank on the GPU, and no torch in the process: the device-driven solve
all-reduces its 16 sums through ncclAllReduce on the solver's stream, the host-side sums and the
in-place array all-reduce of the dictionary update go through the same communicator.  One rank:
every sum is the identity, what runs is the real RCCL call path.  (Two RCCL ranks cannot share a
GPU, and this box has one: N > 1 over RCCL remains unrun here.)"""

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    from sporco_amd import _lib
    _lib.load()
    assert 'torch' not in sys.modules
    from sporco_amd.admm import cbpdn
    from sporco_amd.dist import NativeReducer
    from test_fused_xstep import problem
    red = NativeReducer(0, 1, NativeReducer.unique_id(), device=0)
    assert red.sum([1.5, -2.0, 3.25]) == [1.5, -2.0, 3.25] and red.max(7.0) == 7.0
    assert red.sum(list(range(100))) == [float(v) for v in range(100)]
    D, S = problem(256, 256, 8, 2, seed=4)
    outs = []
    for r in (red, None):
        for optd in ({'MaxMainIter': 8, 'RelStopTol': 0.0}, {'MaxMainIter': 200, 'RelStopTol': 5e-3}):
            b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd), reducer=r)
            Y = b.solve()
            outs.append((Y, b.k, np.asarray(b.getitstat().Rho), np.asarray(b.getitstat().ObjFun)))
    for a, b_ in zip(outs[:2], outs[2:]):
        assert np.array_equal(a[0], b_[0]) and a[1] == b_[1]
        assert np.array_equal(a[2], b_[2]) and np.array_equal(a[3], b_[3])
    assert 3 < outs[1][1] < 200          # (the second pair stopped on the tolerance)
    # host-driven loop (a callback forces it): the sums go through allreduce_host
    calls = []
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 8, 'RelStopTol': 0.0,
                                                           'Callback': lambda o: calls.append(o.k)}),
                       reducer=red)
    Yc = b.solve()
    assert len(calls) == 8 and np.array_equal(Yc, outs[0][0])
    # dictionary learning: the D-step gradient all-reduced in place in device memory
    from sporco_amd.dictlrn import cbpdndl
    rng = np.random.RandomState(9)
    D0, Sd = rng.randn(6, 6, 8).astype(np.float32), rng.randn(256, 256, 2).astype(np.float32)
    res = []
    for r in (red, None):
        opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 4, 'AccurateDFid': True},
                                                xmethod='admm', dmethod='pgm')
        kw = {} if r is None else {'reducer': r}
        d = cbpdndl.ConvBPDNDictLearn(D0, Sd, 0.1, opt, xmethod='admm', dmethod='pgm', **kw)
        res.append((d.solve(), np.asarray(d.getitstat().ObjFun)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    red.close()
    print('NATIVE_RCCL_WORKER_OK')


if __name__ == '__main__':
    main()
