#!/usr/bin/env python3
"""The generic ADMM chain (sizes / dtypes the register kernels do not serve; SPORCO_AMD_UNFUSED=1
forces it at the other sizes): iterations per second in the (Y, U) form (SPORCO_AMD_NO_VFORM=1),
in the single-array form (V = AX + U in place of Y and U, 13 passes instead of 16), with the
column pass as one LDS-resident kernel on top of that (the default where the tile fits: 9 passes).
One JSON line per configuration and variant."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

CONFIGS = [(512, 512, 64, 8, 'float32', True), (384, 384, 64, 8, 'float32', True), (384, 384, 32, 8, 'float32', True),
           (240, 320, 64, 8, 'float32', True), (128, 128, 64, 8, 'float64', False),
           (320, 480, 32, 8, 'float32', True), (256, 256, 32, 8, 'float64', False)]
ONLY = sys.argv[1] if len(sys.argv) > 1 else ''           # substring of '<H>x<W> K=<K>'
VARIANTS = sys.argv[2].split(',') if len(sys.argv) > 2 else ['yu', 'v', 'v_cols_sm']
CONFIGS += [(480, 320, 64, 8, 'float32', True), (256, 256, 64, 8, 'float64', False), (224, 224, 64, 8, 'float32', True)]
for (H, W, K, N, dt, force) in CONFIGS:
    if ONLY not in '%dx%d K=%d %s' % (H, W, K, dt):
        continue
    for variant in VARIANTS:
        env = {'SPORCO_AMD_UNFUSED': '1'} if force else {}
        if variant not in ('v', 'v_cols_sm'):
            env['SPORCO_AMD_NO_VFORM'] = '1'
        if variant != 'v_cols_sm':
            env['SPORCO_AMD_NO_COLS_SM'] = '1'     # (the column pass as three kernels)
        # ('v_cols_sm': one kernel -- the whole tile in LDS, or slabs of filters when it does not fit)
        os.environ.update(env)
        from sporco_amd.admm import cbpdn as ac
        rng = np.random.RandomState(1)
        D = rng.randn(8, 8, K).astype(dt)
        S = rng.randn(H, W, N).astype(dt)
        class NoDownload(ac.ConvBPDN):      # (solve() returns Y: keep the 0.5 GB download out of the timing)
            def getmin(self): return None
        b = NoDownload(D, S, 0.05, ac.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}))
        b.solve(); b._dev.sync()
        b.opt['MaxMainIter'] = 30
        t0 = time.perf_counter(); b.solve(); b._dev.sync()
        dt_s = time.perf_counter() - t0
        b._dev.profile(True)        # (per-kernel times from a second, separately timed run)
        b.solve(); b._dev.sync()
        prof = {k: round(v[0] / max(v[1], 1), 4) for k, v in b._dev.profile_read().items() if v[1]}
        print(json.dumps({'config': '%dx%d K=%d N=%d %s generic chain' % (H, W, K, N, dt),
                          'variant': variant, 'it_per_s': 30 / dt_s, 'kernel_ms': prof}))
        for k in env:
            os.environ.pop(k, None)
        del b
