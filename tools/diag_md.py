"""Diagnostic (GPU): ConvBPDNMaskDcpl fused vs generic chain vs float64 generic at a given shape."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd import _lib
_lib.load()
from sporco_amd.admm import cbpdn
H, K, N = (int(v) for v in (sys.argv[1:4] or (512, 64, 2)))
rng = np.random.RandomState(11)
D = rng.randn(8, 8, K).astype(np.float32)
S = rng.randn(H, H, N).astype(np.float32)
W = (rng.rand(H, H, N) > 0.3).astype(np.float32)
cls = cbpdn.ConvBPDNMaskDcpl
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))
def run(generic, dt=None, iters=5):
    if generic: os.environ['SPORCO_AMD_MD_GENERIC'] = '1'
    else: os.environ.pop('SPORCO_AMD_MD_GENERIC', None)
    o = {'MaxMainIter': iters}
    if dt is not None: o['DataType'] = dt
    b = cls(D, S, 0.1, W, cls.Options(o))
    Y = b.solve()
    return b, Y
for iters in (1, 2, 5):
    bf, Yf = run(False, iters=iters)
    bg, Yg = run(True, iters=iters)
    b64, Y64 = run(True, np.float64, iters=iters)
    out = {'iters': iters, 'Y fused~generic': rel(Yf, Yg), 'Y fused~f64': rel(Yf, Y64), 'Y generic~f64': rel(Yg, Y64),
           'X fused~f64': rel(bf.X, b64.X), 'X generic~f64': rel(bg.X, b64.X),
           'U fused~f64': rel(bf.U, b64.U), 'U generic~f64': rel(bg.U, b64.U),
           'y0 fused~f64': rel(bf.var_y0(), b64.var_y0()), 'y0 generic~f64': rel(bg.var_y0(), b64.var_y0())}
    for f in ('ObjFun', 'DFid', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        out[f + ' fused~f64'] = rel(getattr(bf.getitstat(), f), getattr(b64.getitstat(), f))
        out[f + ' generic~f64'] = rel(getattr(bg.getitstat(), f), getattr(b64.getitstat(), f))
    print(json.dumps(out))
