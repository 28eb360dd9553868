"""Colour dictionary learning at the size of the reference's example scripts, float32:
ConvBPDNDictLearn(dmethod='cns') -- examples/scripts/cdl/cbpdndl_cns_clr-style -- and masked
ConvBPDNMaskDictLearn(xmethod='admm', dmethod='cns') -- examples/scripts/cdl/cbpdndl_md_clr.py:
256 x 256 x 3 images, N = 5, dictionary 8 x 8 x 3 x 32 -- and the multi-scale colour dictionary
of cbpdndl_pgm_clr.py (8/12/16 pixel supports, 96 filters, dmethod='pgm')."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.dictlrn import cbpdndl, cbpdndlmd
rng = np.random.RandomState(3)
H, C, N, M = 256, 3, 5, 32
S = rng.randn(H, H, C, N).astype(np.float32)
D0 = rng.randn(8, 8, C, M).astype(np.float32)
W = (rng.rand(H, H, 1, N) > 0.25).astype(np.float32)
IT = 20


def timed(make, label):
    d = make(3)
    dev = getattr(d.xstep, '_dev', None) or d.xstep.dev
    d.solve(); dev.sync()
    d.opt['MaxMainIter'] = IT
    t0 = time.perf_counter(); d.solve(); dev.sync(); t = time.perf_counter() - t0
    print(json.dumps({'config': label, 'outer_it_per_s': IT / t, 'ms_per_outer_it': 1e3 * t / IT}))


def plain(it):
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': it, 'CCMOD': {'rho': 10.0}},
                                            xmethod='admm', dmethod='cns')
    return cbpdndl.ConvBPDNDictLearn(D0, S, 0.2, opt, xmethod='admm', dmethod='cns')


def masked(it):
    opt = cbpdndlmd.ConvBPDNMaskDictLearn.Options({'MaxMainIter': it, 'CCMOD': {'rho': 1.0}},
                                                  xmethod='admm', dmethod='cns')
    return cbpdndlmd.ConvBPDNMaskDictLearn(D0, S, 0.2, W, opt, xmethod='admm', dmethod='cns')


def multiscale(it):
    Dm = rng.randn(16, 16, C, 96).astype(np.float32)
    dsz = ((8, 8, C, 32), (12, 12, C, 32), (16, 16, C, 32))
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': it, 'DictSize': dsz},
                                            xmethod='pgm', dmethod='pgm')
    return cbpdndl.ConvBPDNDictLearn(Dm, S, 0.2, opt, xmethod='pgm', dmethod='pgm')


timed(plain, "ConvBPDNDictLearn 256x256x3 N=5, dictionary 8x8x3x32, f32, xmethod=admm dmethod=cns")
timed(masked, "ConvBPDNMaskDictLearn 256x256x3 N=5, dictionary 8x8x3x32, f32, xmethod=admm dmethod=cns")
timed(multiscale, "ConvBPDNDictLearn 256x256x3 N=5, multi-scale dictionary 8/12/16 x3 x96, f32, "
                  "xmethod=pgm dmethod=pgm")
