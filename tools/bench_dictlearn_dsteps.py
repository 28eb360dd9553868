"""ConvBPDNDictLearn at 256x256, K=64, N=8, float32 with each dictionary update
(dmethod pgm / cns / ism / cg) and one OnlineConvBPDNDictLearn step per image (256x256, K=64,
100 X-step iterations, the class default)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.dictlrn import cbpdndl, onlinecdl
rng = np.random.RandomState(1)
H, K, N = 256, 64, 8
D0 = rng.randn(8, 8, K).astype(np.float32)
S = rng.randn(H, H, N).astype(np.float32)
IT = 20
for dm in ('pgm', 'cns', 'ism', 'cg'):
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 3}, xmethod='admm', dmethod=dm)
    d = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod=dm)
    d.solve(); d.xstep._dev.sync()
    d.opt['MaxMainIter'] = IT
    t0 = time.perf_counter(); d.solve(); d.xstep._dev.sync(); t = time.perf_counter() - t0
    rec = {'config': "ConvBPDNDictLearn 256x256 K=64 N=8 f32 xmethod=admm dmethod=%s" % dm,
           'outer_it_per_s': IT / t, 'ms_per_outer_it': 1e3 * t / IT}
    if dm == 'cg':
        rec['cg_iterations_last'] = d.dstep.cg_iterations
    print(json.dumps(rec))
cls = onlinecdl.OnlineConvBPDNDictLearn
b = cls(D0, 0.1, cls.Options({'DataType': np.float32}), dimK=0)
b.solve(S[..., 0])
t0 = time.perf_counter()
for i in range(1, 6):
    b.solve(S[..., i])
t = time.perf_counter() - t0
print(json.dumps({'config': "OnlineConvBPDNDictLearn 256x256 K=64 f32, 100 X-step iterations per "
                            "image", 'images_per_s': 5 / t, 'ms_per_image': 1e3 * t / 5}))
