#!/usr/bin/env python3
"""Iteration rates of the masked sparse-coding classes on one GPU: admm.cbpdn.ConvBPDNMaskDcpl and
pgm.cbpdn.ConvBPDNMask at 512x512, K = 64, N = 8, float32 -- on the register-resident kernels
(round 4) and, with SPORCO_AMD_MD_GENERIC=1 / the staged composition, on the generic chain they ran
before -- and AddMaskSim(ConvBPDN) at the same size for comparison.  One JSON line each, with the
per-kernel HIP-event averages of the timed run."""
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
    r = iters / (time.perf_counter() - t0)
    dev.profile(True)
    b.opt['MaxMainIter'] = 5
    b.solve(); dev.sync()
    k = {n: round(v[0] / v[1], 4) for n, v in dev.profile_read().items() if v[1]}
    dev.profile(False)
    return r, k


class StagedMask(pc.ConvBPDNMask):
    def _fused_ok(self):
        return False


for generic in (False, True):
    if generic:
        os.environ['SPORCO_AMD_MD_GENERIC'] = '1'
    b = cbpdn.ConvBPDNMaskDcpl(D, S, 0.05, W, cbpdn.ConvBPDNMaskDcpl.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}))
    r, k = rate(b, b._dev)
    print(json.dumps({'config': 'admm.cbpdn.ConvBPDNMaskDcpl 512x512 K=64 N=8 f32 (%s)'
                                % ('generic chain' if generic else 'register-resident kernels'),
                      'it_per_s': r, 'kernel_ms': k}))
    del b
    os.environ.pop('SPORCO_AMD_MD_GENERIC', None)
    cls = StagedMask if generic else pc.ConvBPDNMask
    b = cls(D, S, 0.05, W, pc.ConvBPDNMask.Options({'MaxMainIter': 3, 'RelStopTol': 0.0, 'L': 500.0}))
    r, k = rate(b, b.dev)
    print(json.dumps({'config': 'pgm.cbpdn.ConvBPDNMask 512x512 K=64 N=8 f32 (%s)'
                                % ('staged composition, generic chain' if generic else 'fused iteration, FLAG_DMASK'),
                      'it_per_s': r, 'kernel_ms': k}))
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
