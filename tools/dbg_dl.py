import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import numpy as np
from conftest import rel_l2, use_backend
use_backend(sys.argv[1] if len(sys.argv)>1 else 'hostsim')
from sporco_amd.dictlrn import cbpdndl
H=W=256; K=4; N=2
rng=np.random.RandomState(3)
D0=rng.randn(5,5,K).astype(np.float32); S=rng.randn(H,W,N).astype(np.float32)
def run(generic):
    if generic: os.environ['SPORCO_AMD_OLD_ROWS']='1'
    try:
        opt=cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter':3,'AccurateDFid':True}, xmethod='admm', dmethod='pgm')
        d=cbpdndl.ConvBPDNDictLearn(D0,S,0.1,opt,xmethod='admm',dmethod='pgm')
    finally:
        os.environ.pop('SPORCO_AMD_OLD_ROWS',None)
    t0=time.time(); D1=d.solve(); t=time.time()-t0
    return d,D1,t
d,D1,t=run(False); d0,D10,t0=run(True)
print('fused rows', d.xstep._dev.uses_fused_rows(), d0.xstep._dev.uses_fused_rows(), 'time %.1f %.1f'%(t,t0))
print('D err', rel_l2(D1,D10), 'coef err', rel_l2(d.getcoef(), d0.getcoef()))
its=d.getitstat(); its0=d0.getitstat()
for f in its._fields:
    if f in ('Iter','Time'): continue
    print('  ',f, rel_l2(np.asarray(getattr(its,f),dtype=float), np.asarray(getattr(its0,f),dtype=float)))
print(' Zf', rel_l2(d.dstep.Zf, d0.dstep.Zf))
