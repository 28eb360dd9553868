#!/usr/bin/env python3
"""Iteration rates of the masked sparse-coding classes on one GPU (generic kernel chain; they have
no register-resident path): admm.cbpdn.ConvBPDNMaskDcpl and pgm.cbpdn.ConvBPDNMask at 512x512,
K = 64, N = 8, float32, and AddMaskSim(ConvBPDN) -- the fused alternative for a boundary / missing-
data mask -- at the same size for comparison.  One JSON line each."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd import _lib
_lib.load()
from sporco_amd.admm import cbpdn
from sporco_amd.pgm import cbpdn as pc

rng = np.random.RandomState(12345)
H, K, N = 512, 64, 8
D = rng.randn(8, 8, K).astype(np.float32)
D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
S = rng.randn(H, H, N).astype(np.float32)
W = (rng.rand(H, H, N) > 0.3).astype(np.float32)


def rate(b, dev, iters=20):
    b._return_min = False
    b.opt['MaxMainIter'] = 3
    b.solve(); dev.sync()
    b.opt['MaxMainIter'] = iters
    t0 = time.perf_counter(); b.solve(); dev.sync()
    return iters / (time.perf_counter() - t0)


b = cbpdn.ConvBPDNMaskDcpl(D, S, 0.05, W, cbpdn.ConvBPDNMaskDcpl.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}))
print(json.dumps({'config': 'admm.cbpdn.ConvBPDNMaskDcpl 512x512 K=64 N=8 f32 (generic chain)',
                  'it_per_s': rate(b, b._dev)}))
del b
b = pc.ConvBPDNMask(D, S, 0.05, W, pc.ConvBPDNMask.Options({'MaxMainIter': 3, 'RelStopTol': 0.0, 'L': 500.0}))
print(json.dumps({'config': 'pgm.cbpdn.ConvBPDNMask 512x512 K=64 N=8 f32 (generic chain)',
                  'it_per_s': rate(b, b.dev)}))
del b
D63 = D[:, :, :63]
a = cbpdn.AddMaskSim(cbpdn.ConvBPDN, D63, S, W, 0.05,
                     opt=cbpdn.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}))
a.cbpdn._return_min = False
a.cbpdn.opt['MaxMainIter'] = 3
a.cbpdn.solve(); a.cbpdn._dev.sync()
a.cbpdn.opt['MaxMainIter'] = 20
t0 = time.perf_counter(); a.cbpdn.solve(); a.cbpdn._dev.sync()
print(json.dumps({'config': 'AddMaskSim(ConvBPDN) 512x512 K=63+1 N=8 f32 (fused kernels, V form)',
                  'it_per_s': 20 / (time.perf_counter() - t0)}))
