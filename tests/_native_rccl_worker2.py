"""Worker of tests/test_dist_native.py::test_two_rank_native_rccl: RANK of WORLD ranks, one GPU each,
the library's own RCCL communicator (no torch in the process; the 128-byte id travels through a
file).  Image shards of unequal size; rank 0 also solves the whole problem alone and compares:
the device-driven ADMM loop (early stop, this rank's host lagging by RANK records), the
host-driven loop, and dictionary learning with the in-place gradient all-reduce."""

import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    from sporco_amd import _lib
    _lib.load()
    assert 'torch' not in sys.modules
    from sporco_amd.admm import cbpdn
    from sporco_amd.dictlrn import cbpdndl
    from sporco_amd.dist import NativeReducer, shard_images
    from test_fused_xstep import problem
    if rank == 0:
        with open(idfile + '.tmp', 'wb') as f:
            f.write(NativeReducer.unique_id())
        os.rename(idfile + '.tmp', idfile)
    for _ in range(600):
        if os.path.exists(idfile):
            break
        time.sleep(0.1)
    uid = open(idfile, 'rb').read()
    red = NativeReducer(rank, world, uid, device=rank)
    assert red.sum([float(rank + 1)])[0] == world * (world + 1) / 2
    n_img = world + 1
    D, S = problem(256, 256, 8, n_img, seed=4)
    kw = dict(device=rank)
    os.environ['SPORCO_AMD_RUN_LAG'] = str(rank)
    b = cbpdn.ConvBPDN(D, shard_images(S, rank, world), 0.05,
                       cbpdn.ConvBPDN.Options({'MaxMainIter': 200, 'RelStopTol': 5e-3}), reducer=red, **kw)
    Y = b.solve()
    os.environ.pop('SPORCO_AMD_RUN_LAG')
    assert red.sum([1.0])[0] == world                        # collectives still aligned
    np.savez(idfile + '.admm.%d.npz' % rank, Y=Y, k=b.k, Rho=np.asarray(b.getitstat().Rho, dtype=float),
             Obj=np.asarray(b.getitstat().ObjFun, dtype=float))
    calls = []
    bc = cbpdn.ConvBPDN(D, shard_images(S, rank, world), 0.05,
                        cbpdn.ConvBPDN.Options({'MaxMainIter': 6, 'RelStopTol': 0.0,
                                                'Callback': lambda o: calls.append(o.k)}), reducer=red, **kw)
    Yc = bc.solve()
    rng = np.random.RandomState(9)
    D0, Sd = rng.randn(6, 6, 8).astype(np.float32), rng.randn(256, 256, n_img).astype(np.float32)
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 4, 'AccurateDFid': True}, xmethod='admm', dmethod='pgm')
    d = cbpdndl.ConvBPDNDictLearn(D0, shard_images(Sd, rank, world), 0.1, opt, xmethod='admm', dmethod='pgm',
                                  reducer=red, **kw)
    D1 = d.solve()
    np.savez(idfile + '.rest.%d.npz' % rank, Yc=Yc, D1=D1, Obj=np.asarray(d.getitstat().ObjFun, dtype=float))
    if rank == 0:
        b1 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 200, 'RelStopTol': 5e-3}), **kw)
        Y1 = b1.solve()
        bc1 = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 6, 'RelStopTol': 0.0,
                                                                  'Callback': lambda o: None}), **kw)
        Yc1 = bc1.solve()
        d1 = cbpdndl.ConvBPDNDictLearn(D0, Sd, 0.1, opt, xmethod='admm', dmethod='pgm', **kw)
        D11 = d1.solve()
        np.savez(idfile + '.single.npz', Y=Y1, k=b1.k, Rho=np.asarray(b1.getitstat().Rho, dtype=float), Yc=Yc1,
                 D1=D11, Obj=np.asarray(d1.getitstat().ObjFun, dtype=float))
    red.sum([0.0])                                            # leave together
    red.close()
    print('NATIVE_RCCL_WORKER2_OK %d' % rank)


if __name__ == '__main__':
    main()
