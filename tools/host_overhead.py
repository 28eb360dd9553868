"""Where the wall-clock of one ADMM iteration goes at config 2: device call vs host bookkeeping."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
from sporco_amd import _lib
rng = np.random.RandomState(1)
K, N, H = 64, 32, 512
D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0, 1), keepdims=True))
S = rng.randn(H, H, N).astype(np.float32)
class R(cbpdn.ConvBPDN):
    def getmin(self): return None
b = R(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 5, 'RelStopTol': 0.0}))
b.solve(); b._dev.sync()
tdev = [0.0]
orig = b._dev.admm_iter
def timed(p):
    t0 = time.perf_counter(); r = orig(p); tdev[0] += time.perf_counter() - t0; return r
b._dev.admm_iter = timed
steps = 40
b.opt['MaxMainIter'] = steps
b.profile(True)
t0 = time.perf_counter(); b.solve(); b._dev.sync(); t = time.perf_counter() - t0
prof = b.profile_read()
kern = sum(v[0] for v in prof.values())
print(json.dumps({'ms_per_it': 1e3 * t / steps, 'device_call_ms': 1e3 * tdev[0] / steps,
                  'host_python_ms': 1e3 * (t - tdev[0]) / steps, 'kernel_sum_ms': kern / steps}))
import cProfile, pstats
b._dev.admm_iter = orig
pr = cProfile.Profile(); pr.enable(); b.solve(); b._dev.sync(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
