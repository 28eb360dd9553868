"""admm.cbpdn.ConvBPDN at 128x128, K=64, N=512, float32 (config-2-sized arrays): the register-resident
kernels (32 x 4 splits) against the generic kernel chain."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(2)
H, K, N = 128, 64, 512
D = rng.randn(8, 8, K).astype(np.float32)
S = rng.randn(H, H, N).astype(np.float32)
class R(cbpdn.ConvBPDN):
    def getmin(self): return None
for unfused in (False, True):
    if unfused: os.environ['SPORCO_AMD_UNFUSED'] = '1'
    try:
        b = R(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}))
    finally:
        os.environ.pop('SPORCO_AMD_UNFUSED', None)
    b.solve(); b._dev.sync()
    b.opt['MaxMainIter'] = 20
    t0 = time.perf_counter(); b.solve(); b._dev.sync(); t = time.perf_counter() - t0
    print(json.dumps({'config': 'admm.cbpdn 128x128 K=64 N=512 f32, default options', 'fused_rows': bool(b._dev.uses_fused_rows()),
                      'it_per_s': 20 / t, 'ms_per_it': 1e3 * t / 20}))
    del b
