import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn
rng = np.random.RandomState(12345)
K, N, H = 64, 4, 512
D = rng.randn(8, 8, K).astype(np.float32); D /= np.sqrt(np.sum(D**2, axis=(0,1), keepdims=True))
S = rng.randn(H, H, N).astype(np.float32)
b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 60, 'RelStopTol': 0.0}))
b.solve()
its = b.getitstat()
print('rho', np.array2string(np.asarray(its.Rho), precision=3, max_line_width=200))
print('r', np.array2string(np.asarray(its.PrimalRsdl), precision=2, max_line_width=200))
