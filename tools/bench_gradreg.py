"""ConvBPDNGradReg at the config-2 shape, next to plain ConvBPDN in the same process."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(1)
K, N, H = 64, 32, 512
D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0,1), keepdims=True))
S = rng.randn(H, H, N).astype(np.float32)
wg = np.zeros(K, np.float32); wg[:4] = 1.0


def run(make, label, steps=40):
    class R(make):
        def getmin(self): return None
    opt = make.Options({'MaxMainIter': 5, 'RelStopTol': 0.0})
    if make is cbpdn.ConvBPDNGradReg:
        opt['GradWeight'] = wg
        b = R(D, S, 0.1, 0.5, opt)
    else:
        b = R(D, S, 0.1, opt)
    b.solve(); b._dev.sync(); b.opt['MaxMainIter'] = 5 + steps
    b.profile(True)
    t0 = time.perf_counter(); b.solve(); b._dev.sync(); t = time.perf_counter() - t0
    prof = {k: round(v[0] / v[1], 4) for k, v in b.profile_read().items() if v[1]}
    print(json.dumps({'config': label, 'it_per_s': steps / t, 'ms_per_it': 1e3 * t / steps,
                      'kernel_ms': prof}))


run(cbpdn.ConvBPDNGradReg, 'ConvBPDNGradReg 512x512 K=64 N=32 f32')
run(cbpdn.ConvBPDN, 'ConvBPDN 512x512 K=64 N=32 f32')
