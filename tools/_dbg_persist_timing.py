import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sporco_amd
here = os.path.dirname(os.path.abspath(__file__))
sporco_amd.load_library(os.path.join(here, 'ubench', 'libsporco_amd_timing.so'))
from sporco_amd.admm import cbpdn as ac
os.environ['SPORCO_AMD_PERSIST_TIMING'] = '1'
rng = np.random.RandomState(12345)
for (H, K, N) in ((256, 32, 1), (128, 64, 1)):
    D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, H, N).astype(np.float32)
    class R1(ac.ConvBPDN):
        def getmin(self): return None
    b = R1(D, S, 0.05, ac.ConvBPDN.Options({'MaxMainIter': 5, 'RelStopTol': 0.0}))
    b.solve(); b._dev.sync()
    b.opt['MaxMainIter'] = 200
    t0 = time.perf_counter(); b.solve(); b._dev.sync()
    print(json.dumps({'shape': [H, K, N], 'it_per_s': 200 / (time.perf_counter() - t0)}))
