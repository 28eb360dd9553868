"""ConvBPDNDictLearn at config 5 (256x256, K=64, N=64, float32) with the ADMM consensus
D-step (dmethod='cns') next to the default PGM D-step."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.dictlrn import cbpdndl
rng = np.random.RandomState(1)
H, K, N = 256, 64, 64
D0 = rng.randn(8, 8, K).astype(np.float32)
S = rng.randn(H, H, N).astype(np.float32)
for dm in ('cns', 'pgm'):
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 5}, xmethod='admm', dmethod=dm)
    d = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod=dm)
    d.solve(); d.xstep._dev.sync()
    d.opt['MaxMainIter'] = 30
    d.xstep.profile(True)
    t0 = time.perf_counter(); d.solve(); d.xstep._dev.sync(); t = time.perf_counter() - t0
    prof = {k: round(v[0] / 30, 4) for k, v in d.xstep.profile_read().items() if v[1]}
    print(json.dumps({'config': "ConvBPDNDictLearn 256x256 K=64 N=64 f32 xmethod=admm dmethod=%s" % dm,
                      'outer_it_per_s': 30 / t, 'ms_per_outer_it': 1e3 * t / 30,
                      'kernel_ms_per_outer_it': prof}))
