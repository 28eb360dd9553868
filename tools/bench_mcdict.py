"""ConvBPDN with a colour dictionary (Cd = 3): 512x512 RGB, K = 64 filters of 8x8x3, N images.
The X-step is the iterated Sherman-Morrison solve (generic kernels)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(1)
K, N, H = 64, int(os.environ.get('N', 16)), 512
D = rng.randn(8, 8, 3, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0, 1, 2), keepdims=True))
S = rng.randn(H, H, 3, N).astype(np.float32)
class R(cbpdn.ConvBPDN):
    def getmin(self): return None
b = R(D, S, 0.1, cbpdn.ConvBPDN.Options({'MaxMainIter': 5, 'RelStopTol': 0.0}))
b.solve(); b._dev.sync(); b.opt['MaxMainIter'] = 20
b.profile(True)
t0 = time.perf_counter(); b.solve(); b._dev.sync(); t = time.perf_counter() - t0
prof = {k: round(v[0] / v[1], 4) for k, v in b.profile_read().items() if v[1]}
E = H * H * N * K * 4
print(json.dumps({'config': 'ConvBPDN RGB dictionary 8x8x3x%d, 512x512x3, N=%d f32' % (K, N),
                  'it_per_s': 20 / t, 'ms_per_it': 50 * t, 'kernel_ms': prof,
                  'ism_solve_GBps': round(2.0 * E * (H // 2 + 1) / (H // 2) / 1e9 / (prof.get('sm_solve', 1) / 1e3), 1)}))
