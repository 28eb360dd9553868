"""AddMaskSim(ConvBPDNGradReg) -- sporco_cuda's cbpdngrdmsk -- at the config-2 image shape with
a 64-filter dictionary (65 filters with the impulse, 66 on the device)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(1)
N, H, K = 32, 512, 64
S = rng.randn(H, H, N).astype(np.float32)
Wm = (rng.rand(H, H, N) > 0.2).astype(np.float32)
D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0, 1), keepdims=True))
wg = np.zeros(K + 1, np.float32); wg[:4] = 1.0
class R(cbpdn.ConvBPDNGradReg):
    def getmin(self): return None
opt = cbpdn.ConvBPDNGradReg.Options({'MaxMainIter': 5, 'RelStopTol': 0.0, 'GradWeight': wg})
b = cbpdn.AddMaskSim(R, D, S, Wm, 0.1, 0.5, opt=opt)
c = b.cbpdn
c.solve(); c._dev.sync(); c.opt['MaxMainIter'] = 30
c.profile(True)
t0 = time.perf_counter(); c.solve(); c._dev.sync(); t = time.perf_counter() - t0
prof = {k: round(v[0] / v[1], 4) for k, v in c.profile_read().items() if v[1]}
print(json.dumps({'config': 'AddMaskSim(ConvBPDNGradReg) 512x512 K=64+1 N=32 f32',
                  'fused_rows': bool(c._dev.uses_fused_rows()), 'it_per_s': 30 / t,
                  'ms_per_it': 1e3 * t / 30, 'kernel_ms': prof}))
