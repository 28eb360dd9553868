import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(1)
K, N, H = 128, 16, 512
D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0,1), keepdims=True))
S = rng.randn(H, H, N).astype(np.float32)
class R(cbpdn.ConvBPDN):
    def getmin(self): return None
b = R(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}))
b.solve(); b._dev.sync(); b.opt['MaxMainIter'] = 20
t0 = time.perf_counter(); b.solve(); b._dev.sync(); t = time.perf_counter() - t0
print(json.dumps({'config': 'admm.cbpdn 512x512 K=128 N=16 f32 (same array sizes as config 2)', 'fused': b._dev.uses_fused_rows(), 'it_per_s': 20 / t, 'ms_per_it': 50 * t}))
