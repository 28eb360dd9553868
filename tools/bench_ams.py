"""AddMaskSim(ConvBPDN) at the config-2 image shape: K = 63 (+ impulse = 64 filters, the
fused kernels) and K = 64 (+ impulse = 65 filters)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(1)
N, H = 32, 512
S = rng.randn(H, H, N).astype(np.float32)
Wm = (rng.rand(H, H, N) > 0.2).astype(np.float32)


def run(K, steps=30):
    D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0, 1), keepdims=True))
    class R(cbpdn.ConvBPDN):
        def getmin(self): return None
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 5, 'RelStopTol': 0.0})
    b = cbpdn.AddMaskSim(R, D, S, Wm, 0.1, opt=opt)
    c = b.cbpdn
    c.solve(); c._dev.sync(); c.opt['MaxMainIter'] = 5 + steps
    c.profile(True)
    t0 = time.perf_counter(); c.solve(); c._dev.sync(); t = time.perf_counter() - t0
    prof = {k: round(v[0] / v[1], 4) for k, v in c.profile_read().items() if v[1]}
    print(json.dumps({'config': 'AddMaskSim(ConvBPDN) 512x512 K=%d+1 N=32 f32' % K,
                      'fused_rows': bool(c._dev.uses_fused_rows()),
                      'it_per_s': steps / t, 'ms_per_it': 1e3 * t / steps, 'kernel_ms': prof}))


for K in [int(v) for v in os.environ.get('KS', '63,64').split(',')]:
    run(K)
