"""Worker of tests/test_dist_gloo.py: one rank of a 2-process image-sharded solve.

Runs on CPU: gloo process group + the fiber-simulator build of the kernels."""

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
    from conftest import HOSTSIM_LIB, load_golden
    sporco_amd.load_library(HOSTSIM_LIB)
    from sporco_amd.admm import cbpdn
    from sporco_amd.dist import TorchReducer, shard_images
    g = load_golden('admm_multichan_f64')          # S: (16, 12, 3, 2): two images
    S = shard_images(g['S'], rank, world, axis=-1)
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 25})
    b = cbpdn.ConvBPDN(g['D'], S, float(g['lmbda']), opt, reducer=TorchReducer())
    Y = b.solve()
    its = b.getitstat()
    np.savez(out_path + '.%d.npz' % rank, Y=Y, ObjFun=np.array(its.ObjFun),
             Rho=np.array(its.Rho), PrimalRsdl=np.array(its.PrimalRsdl),
             DualRsdl=np.array(its.DualRsdl), k=b.k)
    # the staged path (a step method overridden => one device call per step) under sharding:
    # the X-step's own sums (data fidelity) are all-reduced like the residual sums

    class Hooked(cbpdn.ConvBPDN):
        def ystep(self):
            super(Hooked, self).ystep()
    h = Hooked(g['D'], S, float(g['lmbda']), cbpdn.ConvBPDN.Options({'MaxMainIter': 25}),
               reducer=TorchReducer())
    assert not h._fused_ok()
    Yh = h.solve()
    its = h.getitstat()
    np.savez(out_path + '.hook.%d.npz' % rank, Y=Yh, ObjFun=np.array(its.ObjFun),
             DFid=np.array(its.DFid), Rho=np.array(its.Rho), k=h.k)
    # the device-driven loop (three-launch float32 path) under sharding: one image per rank,
    # the all-reduce hooked in between the local sums and the device-side control update
    # (128 x 128, K = 130: the 32 x 4 splits and the cooperating slab workgroups of the K > 64
    # column pass run under the hook as well)
    rng = np.random.RandomState(99)
    Df = rng.randn(4, 4, 130).astype(np.float32)
    Df /= np.sqrt(np.sum(Df ** 2, axis=(0, 1), keepdims=True))
    Sf = rng.randn(128, 128, 2).astype(np.float32)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0}
    bd = cbpdn.ConvBPDN(Df, shard_images(Sf, rank, world, axis=-1), 0.05,
                        cbpdn.ConvBPDN.Options(optd), reducer=TorchReducer())
    assert bd._device_loop_ok() and bd._dev.uses_fused_rows()
    assert bd._reducer.device_sum_hook(bd._dev) is not None
    Yd = bd.solve()
    its = bd.getitstat()
    out = dict(Y=Yd, k=bd.k, **{f: np.asarray(getattr(its, f), dtype=float)
                                for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho')})
    if rank == 0:       # the single-process run of both images, host-driven loop
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
        b1 = cbpdn.ConvBPDN(Df, Sf, 0.05, cbpdn.ConvBPDN.Options(optd))
        out['Y_single'] = b1.solve()
        i1 = b1.getitstat()
        out.update({f + '_single': np.asarray(getattr(i1, f), dtype=float)
                    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho')})
        os.environ.pop('SPORCO_AMD_HOST_LOOP')
    np.savez(out_path + '.devloop.%d.npz' % rank, **out)
    # the same loop stopping EARLY on its tolerance test, with the ranks' hosts noticing the
    # stop at different times (rank 1 does not see its newest three records): every rank must
    # still issue the same number of all-reduces, or the collective after the solve pairs
    # with a surplus one (a hang or a corrupted sum over RCCL)
    Ds = Df[:, :, :8].copy()
    opts = {'MaxMainIter': 60, 'RelStopTol': 2e-2}
    os.environ['SPORCO_AMD_RUN_LAG'] = '3' if rank == 1 else '0'
    red = TorchReducer()
    be = cbpdn.ConvBPDN(Ds, shard_images(Sf, rank, world, axis=-1), 0.05,
                        cbpdn.ConvBPDN.Options(opts), reducer=red)
    assert be._device_loop_ok() and be._reducer.device_sum_hook(be._dev) is not None
    Ye = be.solve()
    os.environ.pop('SPORCO_AMD_RUN_LAG')
    after = red.sum([float(rank + 1)])[0]          # 3.0 only if the collectives are aligned
    ite = be.getitstat()
    oute = dict(Y=Ye, k=be.k, after=after, Rho=np.asarray(ite.Rho, dtype=float),
                PrimalRsdl=np.asarray(ite.PrimalRsdl, dtype=float))
    if rank == 0:
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
        b1 = cbpdn.ConvBPDN(Ds, Sf, 0.05, cbpdn.ConvBPDN.Options(opts))
        oute['Y_single'] = b1.solve()
        oute['k_single'] = b1.k
        os.environ.pop('SPORCO_AMD_HOST_LOOP')
    np.savez(out_path + '.earlystop.%d.npz' % rank, **oute)
    # dictionary learning, four images over the two ranks: X-step sums and the D-step
    # gradient are all-reduced, the dictionary is replicated
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden('cbpdndl_shard_f64')
    S = shard_images(g['S'], rank, world, axis=-1)
    opt = cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 10, 'AccurateDFid': True, 'CCMOD': {'ZeroMean': True}},
        xmethod='admm', dmethod='pgm')
    d = cbpdndl.ConvBPDNDictLearn(g['D0'], S, float(g['lmbda']), opt, xmethod='admm',
                                  dmethod='pgm', reducer=TorchReducer())
    D1 = d.solve()
    its = d.getitstat()
    np.savez(out_path + '.dl.%d.npz' % rank, D1=D1, X=d.getcoef(),
             **{f: np.asarray(getattr(its, f), dtype=float) for f in its._fields
                if f not in ('Iter', 'Time')})
    # FISTA sparse coding with one image per rank: backtracking and Barzilai-Borwein step sizes
    # take their inner products over all images
    from sporco_amd.pgm import cbpdn as pgm_cbpdn
    from test_pgm_cbpdn import policies
    for name in ('pgm_btstd_f64', 'pgm_stepbb_f64'):
        g = load_golden(name)
        optd = dict(policies()[name])
        optd['RelStopTol'] = 0.0
        S = shard_images(g['S'], rank, world, axis=-1)
        b = pgm_cbpdn.ConvBPDN(g['D'], S, float(g['lmbda']), pgm_cbpdn.ConvBPDN.Options(optd),
                               reducer=TorchReducer())
        X = b.solve()
        its = b.getitstat()
        np.savez(out_path + '.%s.%d.npz' % (name, rank), X=X, k=b.k,
                 **{f: np.asarray(getattr(its, f), dtype=float) for f in ('ObjFun', 'Rsdl', 'L')})
    # ... and dictionary learning with that X-step
    g = load_golden('cbpdndl_shard_f64')
    S = shard_images(g['S'], rank, world, axis=-1)
    outs = []
    for red in (TorchReducer(), None):
        opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 4, 'AccurateDFid': True},
                                                xmethod='pgm', dmethod='pgm')
        kw = {'reducer': red} if red is not None else {}
        d = cbpdndl.ConvBPDNDictLearn(g['D0'], S if red is not None else g['S'],
                                      float(g['lmbda']), opt, xmethod='pgm', dmethod='pgm', **kw)
        outs.append((d.solve(), np.asarray(d.getitstat().ObjFun, dtype=float)))
    np.savez(out_path + '.dlpgm.%d.npz' % rank, D1=outs[0][0], D1_single=outs[1][0],
             ObjFun=outs[0][1], ObjFun_single=outs[1][1])
    # mask decoupling: the X-step alone (one image and its mask per rank), then masked dictionary
    # learning with either X-step and the masked PGM D-step
    g = load_golden('maskdcpl_f64')
    b = cbpdn.ConvBPDNMaskDcpl(g['D'], shard_images(g['S'], rank, world),
                               float(g['lmbda']), shard_images(g['W'], rank, world),
                               cbpdn.ConvBPDNMaskDcpl.Options({'MaxMainIter': 30}),
                               reducer=TorchReducer())
    Y1 = b.solve()
    its = b.getitstat()
    np.savez(out_path + '.mdcpl.%d.npz' % rank, Y1=Y1, k=b.k,
             **{f: np.asarray(getattr(its, f), dtype=float)
                for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual')})
    from sporco_amd.dictlrn import cbpdndlmd
    for xm in ('admm', 'pgm'):
        g = load_golden('cbpdndlmd_shard_%s_f64' % xm)
        opt = cbpdndlmd.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                                      xmethod=xm, dmethod='pgm')
        d = cbpdndlmd.ConvBPDNMaskDictLearn(
            g['D0'], shard_images(g['S'], rank, world), float(g['lmbda']),
            shard_images(g['W'], rank, world), opt, xmethod=xm, dmethod='pgm',
            reducer=TorchReducer())
        D1 = d.solve()
        its = d.getitstat()
        np.savez(out_path + '.dlmd_%s.%d.npz' % (xm, rank), D1=D1, X=d.getcoef(),
                 **{f: np.asarray(getattr(its, f), dtype=float)
                    for f in ('ObjFun', 'DFid', 'RegL1')})
    # ADMM consensus dictionary updates, images sharded: the consensus average is the one array
    # all-reduce per iteration; standalone (plain and mask decoupling), then inside dictionary
    # learning with the sharded ADMM X-steps
    from sporco_amd.admm import ccmod as admm_ccmod
    from sporco_amd.admm import ccmodmd as admm_ccmodmd
    autorho = {'Enabled': True, 'Period': 3, 'Scaling': 2.0, 'RsdlRatio': 1.2,
               'AutoScaling': True, 'RsdlTarget': 1.0}
    optd = {'MaxMainIter': 15, 'ZeroMean': True, 'AutoRho': autorho}
    for name, masked in (('ccmod_cns_shard_f64', False), ('ccmodmd_cns_shard_f64', True)):
        g = load_golden(name)
        dsz = tuple(int(v) for v in g['dsz'])
        Zs, Ss = shard_images(g['Z'], rank, world, axis=3), shard_images(g['S'], rank, world)
        o = admm_ccmod.ConvCnstrMOD_Consensus.Options(optd)
        if masked:
            c = admm_ccmodmd.ConvCnstrMODMaskDcpl_Consensus(
                Zs, Ss, shard_images(g['W'], rank, world), dsz, o, reducer=TorchReducer())
        else:
            c = admm_ccmod.ConvCnstrMOD_Consensus(Zs, Ss, dsz, o, reducer=TorchReducer())
        c.solve()
        its = c.getitstat()
        np.savez(out_path + '.%s.%d.npz' % (name, rank), D=c.getdict(), Y=c.Y, k=c.k,
                 **{f: np.asarray(getattr(its, f), dtype=float)
                    for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')})
    g = load_golden('cbpdndl_shard_cns_f64')
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                            xmethod='admm', dmethod='cns')
    d = cbpdndl.ConvBPDNDictLearn(g['D0'], shard_images(g['S'], rank, world), float(g['lmbda']),
                                  opt, xmethod='admm', dmethod='cns', reducer=TorchReducer())
    D1 = d.solve()
    its = d.getitstat()
    np.savez(out_path + '.dlcns.%d.npz' % rank, D1=D1, X=d.getcoef(),
             **{f: np.asarray(getattr(its, f), dtype=float) for f in its._fields
                if f not in ('Iter', 'Time')})
    g = load_golden('cbpdndlmd_shard_cns_f64')
    opt = cbpdndlmd.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                                  xmethod='admm', dmethod='cns')
    d = cbpdndlmd.ConvBPDNMaskDictLearn(
        g['D0'], shard_images(g['S'], rank, world), float(g['lmbda']),
        shard_images(g['W'], rank, world), opt, xmethod='admm', dmethod='cns',
        reducer=TorchReducer())
    D1 = d.solve()
    its = d.getitstat()
    np.savez(out_path + '.dlmdcns.%d.npz' % rank, D1=D1, X=d.getcoef(),
             **{f: np.asarray(getattr(its, f), dtype=float) for f in its._fields
                if f not in ('Iter', 'Time')})
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
