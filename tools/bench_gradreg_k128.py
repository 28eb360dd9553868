"""ConvBPDNGradReg with 128 filters (slab column kernels), config-2-sized arrays."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(1)
K, N, H = 128, 16, 512
D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0, 1), keepdims=True))
S = rng.randn(H, H, N).astype(np.float32)
wg = np.zeros(K, np.float32); wg[:4] = 1.0
class R(cbpdn.ConvBPDNGradReg):
    def getmin(self): return None
b = R(D, S, 0.1, 0.5, cbpdn.ConvBPDNGradReg.Options({'MaxMainIter': 5, 'RelStopTol': 0.0, 'GradWeight': wg}))
b.solve(); b._dev.sync(); b.opt['MaxMainIter'] = 30
b.profile(True)
t0 = time.perf_counter(); b.solve(); b._dev.sync(); t = time.perf_counter() - t0
prof = {k: round(v[0] / v[1], 4) for k, v in b.profile_read().items() if v[1]}
print(json.dumps({'config': 'ConvBPDNGradReg 512x512 K=128 N=16 f32', 'fused_rows': bool(b._dev.uses_fused_rows()),
                  'it_per_s': 30 / t, 'ms_per_it': 1e3 * t / 30, 'kernel_ms': prof}))
