"""ConvBPDNDictLearn 256x256, K=128, N=32, float32 (xmethod admm, dmethod pgm): the tile-major
setcoef / gradient for K > 64 against the generic chain."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.dictlrn import cbpdndl
rng = np.random.RandomState(1)
H, K, N = 256, 128, 32
D0 = rng.randn(8, 8, K).astype(np.float32)
S = rng.randn(H, H, N).astype(np.float32)
for generic in (False, True):
    if generic:
        os.environ['SPORCO_AMD_OLD_ROWS'] = '1'; os.environ['SPORCO_AMD_NO_PAD'] = '1'
    try:
        opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 3}, xmethod='admm', dmethod='pgm')
        d = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod='pgm')
    finally:
        os.environ.pop('SPORCO_AMD_OLD_ROWS', None); os.environ.pop('SPORCO_AMD_NO_PAD', None)
    d.solve(); d.xstep._dev.sync()
    d.opt['MaxMainIter'] = 20
    t0 = time.perf_counter(); d.solve(); d.xstep._dev.sync(); t = time.perf_counter() - t0
    print(json.dumps({'config': 'ConvBPDNDictLearn 256x256 K=128 N=32 f32 (admm X / pgm D)',
                      'fused_rows': bool(d.xstep._dev.uses_fused_rows()), 'outer_it_per_s': 20 / t,
                      'ms_per_outer_it': 1e3 * t / 20}))
    del d
