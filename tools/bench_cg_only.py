"""ConvBPDNDictLearn 256x256 K=64 N=8 float32 with the CG dictionary update only (for profiles)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.dictlrn import cbpdndl
rng = np.random.RandomState(1)
H, K, N = 256, 64, 8
D0 = rng.randn(8, 8, K).astype(np.float32)
S = rng.randn(H, H, N).astype(np.float32)
IT = 20
opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 3}, xmethod='admm', dmethod='cg')
d = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod='cg')
d.solve(); d.xstep._dev.sync()
d.opt['MaxMainIter'] = IT
t0 = time.perf_counter(); d.solve(); d.xstep._dev.sync(); t = time.perf_counter() - t0
print(json.dumps({'dmethod': 'cg', 'outer_it_per_s': IT / t, 'cg_iterations_last': d.dstep.cg_iterations}))
