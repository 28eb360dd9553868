"""pgm.cbpdn.ConvBPDN at 512x512, K=128, N=16 (config-4-sized arrays), float32: the fused iteration
(cooperating slab workgroups in the gradient step) against the staged composition."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.pgm import cbpdn as pc
rng = np.random.RandomState(2)
H, K, N = 512, 128, 16
D = rng.randn(8, 8, K).astype(np.float32)
S = rng.randn(H, H, N).astype(np.float32)
class R(pc.ConvBPDN):
    def getmin(self): return None
for generic in (False, True):
    if generic:
        os.environ['SPORCO_AMD_OLD_ROWS'] = '1'; os.environ['SPORCO_AMD_NO_PAD'] = '1'
    try:
        b = R(D, S, 0.05, pc.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0, 'L': 500.0}))
    finally:
        os.environ.pop('SPORCO_AMD_OLD_ROWS', None); os.environ.pop('SPORCO_AMD_NO_PAD', None)
    b.solve(); b.dev.sync()
    b.opt['MaxMainIter'] = 20
    t0 = time.perf_counter(); b.solve(); b.dev.sync(); t = time.perf_counter() - t0
    print(json.dumps({'config': 'pgm.cbpdn 512x512 K=128 N=16 f32', 'fused': bool(b._fused_ok()),
                      'it_per_s': 20 / t, 'ms_per_it': 1e3 * t / 20}))
    del b
