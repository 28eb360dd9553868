"""ConvBPDNDictLearn with a colour dictionary (8x8x3x64) on 256x256 RGB images."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.dictlrn import cbpdndl
rng = np.random.RandomState(1)
H, K, N = 256, 64, int(os.environ.get('N', 32))
D0 = rng.randn(8, 8, 3, K).astype(np.float32)
S = rng.randn(H, H, 3, N).astype(np.float32)
opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 5}, xmethod='admm', dmethod='pgm')
d = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod='pgm')
d.solve(); d.xstep._dev.sync()
d.opt['MaxMainIter'] = 30
d.xstep.profile(True)
t0 = time.perf_counter(); d.solve(); d.xstep._dev.sync(); t = time.perf_counter() - t0
prof = {k: round(v[0] / 30, 4) for k, v in d.xstep.profile_read().items() if v[1]}
print(json.dumps({'config': 'ConvBPDNDictLearn RGB dictionary 8x8x3x%d, 256x256x3, N=%d f32' % (K, N),
                  'fused_xstep': bool(d.xstep._dev.uses_fused_rows()),
                  'outer_it_per_s': 30 / t, 'ms_per_outer_it': 1e3 * t / 30,
                  'kernel_ms_per_outer_it': prof}))
