#!/usr/bin/env python3
"""Mixed-radix image sizes (H, W = 16 x {10, 12, 14, 15, 18, 20, 21, 24, 25, 27, 28, 30}: csc_rows_mr.hip, csc_fused.h) on the
register-resident kernels against the generic chain (SPORCO_AMD_UNFUSED=1) of the same library:
iterations/s, per-kernel times, and the agreement of the two after a few iterations.
One JSON line per shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sporco_amd.admm import cbpdn as ac

SHAPES = [(384, 384, 32, 8), (480, 320, 64, 8), (240, 320, 64, 8), (320, 480, 32, 8), (448, 448, 64, 8),
          (384, 512, 64, 8), (480, 480, 64, 16), (224, 224, 64, 8), (160, 192, 64, 16), (336, 400, 32, 8),
          (432, 288, 64, 8)]
ONLY = sys.argv[1] if len(sys.argv) > 1 else ''


class NoDownload(ac.ConvBPDN):
    def getmin(self):
        return None


def solver(D, S, unfused, iters, cls=ac.ConvBPDN):
    if unfused:
        os.environ['SPORCO_AMD_UNFUSED'] = '1'
    try:
        return cls(D, S, 0.05, cls.Options({'MaxMainIter': iters, 'RelStopTol': 0.0}))
    finally:
        os.environ.pop('SPORCO_AMD_UNFUSED', None)


for (H, W, K, N) in SHAPES:
    if ONLY == "none":
        break
    if ONLY not in '%dx%d K=%d' % (H, W, K):
        continue
    rng = np.random.RandomState(1)
    D = rng.randn(8, 8, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, N).astype(np.float32)
    out = {'config': 'admm.cbpdn.ConvBPDN %dx%d K=%d N=%d float32, default options' % (H, W, K, N)}
    ys = {}
    for name, unfused in (('register', False), ('generic', True)):
        b = solver(D, S, unfused, 6)
        ys[name] = b.solve().astype(np.float64)
        tr = np.asarray(b.getitstat().ObjFun, float)
        ys[name + '_obj'] = tr
        out[name + '_engaged'] = bool(b._dev.uses_fused_rows() and b._dev.uses_fused_cols())
        del b
        b = solver(D, S, unfused, 5, NoDownload)
        b.solve(); b._dev.sync()
        b.opt['MaxMainIter'] = 60
        t0 = time.perf_counter(); b.solve(); b._dev.sync()
        out[name + '_it_per_s'] = 60 / (time.perf_counter() - t0)
        b._dev.profile(True)
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
        b.opt['MaxMainIter'] = 20
        b.solve(); b._dev.sync()
        os.environ.pop('SPORCO_AMD_HOST_LOOP')
        out[name + '_kernel_ms'] = {k: round(v[0] / max(v[1], 1), 4) for k, v in b._dev.profile_read().items() if v[1]}
        b._dev.profile(False)
        del b
    out['speedup'] = out['register_it_per_s'] / out['generic_it_per_s']
    out['rel_l2_Y_register_vs_generic'] = float(np.linalg.norm(ys['register'] - ys['generic']) / np.linalg.norm(ys['generic']))
    out['ObjFun_max_rel_dev'] = float(np.max(np.abs(ys['register_obj'] - ys['generic_obj']) / np.abs(ys['generic_obj'])))
    print(json.dumps(out), flush=True)


# ---- FISTA and dictionary learning at mixed-radix sizes (csc_pgm_mr.hip) --------------------------
def fista_and_cdl():
    from sporco_amd.pgm import cbpdn as pc
    from sporco_amd.pgm.backtrack import BacktrackStandard
    from sporco_amd.dictlrn import cbpdndl
    for (H, W, K, N) in [(384, 384, 32, 8), (480, 320, 64, 8), (240, 320, 64, 16)]:
        if ONLY not in '%dx%d K=%d' % (H, W, K):
            continue
        rng = np.random.RandomState(2)
        D = rng.randn(8, 8, K).astype(np.float32)
        D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
        S = rng.randn(H, W, N).astype(np.float32)
        out = {'config': 'pgm.cbpdn.ConvBPDN and ConvBPDNDictLearn(admm, pgm) %dx%d K=%d N=%d float32' % (H, W, K, N)}
        for name, unfused in (('register', False), ('generic', True)):
            for tag, extra in (('fista', {}), ('fista_bt', {'Backtrack': BacktrackStandard()})):
                if unfused:
                    os.environ['SPORCO_AMD_UNFUSED'] = '1'
                try:
                    b = pc.ConvBPDN(D, S, 0.05, pc.ConvBPDN.Options(dict({'MaxMainIter': 5, 'RelStopTol': 0.0, 'L': 500.0}, **extra)))
                finally:
                    os.environ.pop('SPORCO_AMD_UNFUSED', None)
                b._return_min = False
                b.solve(); b.dev.sync()
                b.opt['MaxMainIter'] = 40
                t0 = time.perf_counter(); b.solve(); b.dev.sync()
                out['%s_%s_it_per_s' % (name, tag)] = 40 / (time.perf_counter() - t0)
                out['%s_%s_objfun' % (name, tag)] = float(b.getitstat().ObjFun[-1])
                out['%s_%s_fused' % (name, tag)] = bool(b._fused_ok())
                del b
            if unfused:
                os.environ['SPORCO_AMD_UNFUSED'] = '1'
            try:
                opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 5}, xmethod='admm', dmethod='pgm')
                d = cbpdndl.ConvBPDNDictLearn(D, S, 0.1, opt, xmethod='admm', dmethod='pgm')
            finally:
                os.environ.pop('SPORCO_AMD_UNFUSED', None)
            d.solve(); d.xstep._dev.sync()
            d.opt['MaxMainIter'] = 30
            t0 = time.perf_counter(); d.solve(); d.xstep._dev.sync()
            out[name + '_cdl_outer_it_per_s'] = 30 / (time.perf_counter() - t0)
            out[name + '_cdl_objfun'] = float(d.getitstat().ObjFun[-1])
            del d
        print(json.dumps(out), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == 'pgm':
    fista_and_cdl()


# ---- mask decoupling at mixed-radix sizes (api_maskdcpl.inc on csc_rows_mr / csc_fused) ----------
def mask_decoupling():
    for (H, W, K, N) in [(480, 320, 64, 8), (384, 384, 32, 8), (240, 320, 64, 8)]:
        rng = np.random.RandomState(2)
        D = rng.randn(8, 8, K).astype(np.float32)
        D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
        S = rng.randn(H, W, N).astype(np.float32)
        M = (rng.rand(H, W, N) > 0.3).astype(np.float32)
        out = {'config': 'admm.cbpdn.ConvBPDNMaskDcpl %dx%d K=%d N=%d float32, default options' % (H, W, K, N)}
        ys = {}
        for name, generic in (('register', False), ('generic', True)):
            if generic:
                os.environ['SPORCO_AMD_MD_GENERIC'] = '1'
            try:
                cls = ac.ConvBPDNMaskDcpl
                b = cls(D, S, 0.05, M, cls.Options({'MaxMainIter': 5, 'RelStopTol': 0.0}))
                b._return_min = False
                b.solve(); b._dev.sync()
                ys[name] = np.asarray(b.Y, np.float64)
                b.opt['MaxMainIter'] = 40
                t0 = time.perf_counter(); b.solve(); b._dev.sync()
                out[name + '_it_per_s'] = 40 / (time.perf_counter() - t0)
                b._dev.profile(True)
                b.opt['MaxMainIter'] = 10
                b.solve(); b._dev.sync()
                out[name + '_kernel_ms'] = {k: round(v[0] / max(v[1], 1), 4) for k, v in b._dev.profile_read().items() if v[1]}
                b._dev.profile(False)
                del b
            finally:
                os.environ.pop('SPORCO_AMD_MD_GENERIC', None)
        out['speedup'] = out['register_it_per_s'] / out['generic_it_per_s']
        out['rel_l2_Y_register_vs_generic'] = float(np.linalg.norm(ys['register'] - ys['generic']) / np.linalg.norm(ys['generic']))
        print(json.dumps(out), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == 'md':
    mask_decoupling()
