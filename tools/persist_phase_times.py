#!/usr/bin/env python3
"""Phase times of the one-launch solve of small problems (csc_rows.hip admm_persist_kernel) from a
MEASUREMENT build of the library: compile csc_rows.hip with -DSA_PERSIST_TIMING and link it with
the product's other objects into tools/ubench/libsporco_amd_timing.so, e.g.

    cd sporco_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -fno-slp-vectorize \\
        -DSA_PERSIST_TIMING -c csc_rows.hip -o /tmp/csc_rows_timing.o && \\
    hipcc --offload-arch=gfx950 -shared -fPIC fft.o csc_kernels.o csc_fused.o csc_fused_mc.o \\
        /tmp/csc_rows_timing.o csc_pgm.o csc_api.o -o ../../tools/ubench/libsporco_amd_timing.so

Prints, per solve, the average ticks (10 ns) workgroup 0 spends in each phase
(profiles/r03_persist.md)."""
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
