"""The 8-wave column kernels whose register count decides between one and two workgroups per CU:
ConvBPDNGradReg at 256 x 256, K = 64, and ConvBPDN at 256 x 256, K = 128 (per-kernel averages)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(1)


def run(make, label, H, K, N, steps=60):
    D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, H, N).astype(np.float32)

    class R(make):
        def getmin(self): return None
    opt = make.Options({'MaxMainIter': 5, 'RelStopTol': 0.0, 'AutoRho': {'Enabled': False}, 'rho': 5.0})
    if make is cbpdn.ConvBPDNGradReg:
        wg = np.zeros(K, np.float32); wg[:4] = 1.0
        opt['GradWeight'] = wg
        b = R(D, S, 0.1, 0.5, opt)
    else:
        b = R(D, S, 0.1, opt)
    b.solve(); b._dev.sync(); b.opt['MaxMainIter'] = 5 + steps
    b.profile(True)
    t0 = time.perf_counter(); b.solve(); b._dev.sync(); t = time.perf_counter() - t0
    prof = {k: round(v[0] / v[1], 4) for k, v in b.profile_read().items() if v[1]}
    print(json.dumps({'config': label, 'library': os.environ.get('SPORCO_AMD_LIBRARY', 'product'),
                      'it_per_s': steps / t, 'ms_per_it': 1e3 * t / steps, 'kernel_ms': prof}))


run(cbpdn.ConvBPDNGradReg, 'ConvBPDNGradReg 256x256 K=64 N=32 f32', 256, 64, 32)
run(cbpdn.ConvBPDN, 'ConvBPDN 256x256 K=128 N=16 f32', 256, 128, 16)
run(cbpdn.ConvBPDN, 'ConvBPDN 128x128 K=128 N=32 f32', 128, 128, 32)
