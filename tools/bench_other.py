#!/usr/bin/env python3
"""Iteration rates of the other BASELINE.json configurations on one GPU
(config 4: pgm.cbpdn FISTA 512x512 K=64 N=32; config 5: ConvBPDNDictLearn 256x256 K=64 N=64;
config 1: 256x256 K=32 N=1).  Prints one JSON line per configuration."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sporco_amd
from sporco_amd import _lib
_lib.load()

def problem(H, W, K, N, seed=12345):
    rng = np.random.RandomState(seed)
    D = rng.randn(8, 8, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, N).astype(np.float32)
    return D, S

def timed_solve(b, iters, dev):
    b.opt['MaxMainIter'] = 3
    b.solve(); dev.sync()
    b.opt['MaxMainIter'] = iters
    t0 = time.perf_counter(); b.solve(); dev.sync()
    return iters / (time.perf_counter() - t0)

which = sys.argv[1:] or ['pgm', 'dl', 'c1']
if 'pgm' in which:
    from sporco_amd.pgm import cbpdn as pc
    D, S = problem(512, 512, 64, 32)
    class R(pc.ConvBPDN):
        def getmin(self): return None
    for bt in (False, True):
        optd = {'MaxMainIter': 3, 'RelStopTol': 0.0, 'L': 500.0}
        if bt:
            from sporco_amd.pgm.backtrack import BacktrackStandard
            optd['Backtrack'] = BacktrackStandard()
        b = R(D, S, 0.05, pc.ConvBPDN.Options(optd))
        r = timed_solve(b, 20, b.dev)
        print(json.dumps({'config': 'pgm.cbpdn 512x512 K=64 N=32 f32' + (' BacktrackStandard' if bt else ''), 'it_per_s': r}))
        del b
if 'dl' in which:
    from sporco_amd.dictlrn import cbpdndl
    D, S = problem(256, 256, 64, 64)
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 3}, xmethod='admm', dmethod='pgm')
    d = cbpdndl.ConvBPDNDictLearn(D, S, 0.1, opt, xmethod='admm', dmethod='pgm')
    d.solve(); d.xstep._dev.sync()
    d.opt['MaxMainIter'] = 20
    t0 = time.perf_counter(); d.solve(); d.xstep._dev.sync()
    print(json.dumps({'config': 'ConvBPDNDictLearn 256x256 K=64 N=64 f32 (admm X / pgm D)', 'outer_it_per_s': 20 / (time.perf_counter() - t0)}))
    del d
if 'c1' in which:
    from sporco_amd.admm import cbpdn as ac
    D, S = problem(256, 256, 32, 1)
    class R1(ac.ConvBPDN):
        def getmin(self): return None
    b = R1(D, S[:, :, 0], 0.05, ac.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}), dimK=0)
    print(json.dumps({'config': 'admm.cbpdn 256x256 K=32 N=1 f32 default options', 'it_per_s': timed_solve(b, 200, b._dev)}))
