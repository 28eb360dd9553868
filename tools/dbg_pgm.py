import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import numpy as np
from conftest import rel_l2, use_backend
use_backend(sys.argv[1] if len(sys.argv)>1 else 'hostsim')
from oracle import cbpdn_oracle as orc
from sporco_amd.pgm import cbpdn as pc
from test_fused_xstep import problem
for (H,W,K,N,it) in [(256,256,4,1,4),(256,512,6,1,3)]:
    D,S = problem(H,W,K,N,seed=H+K)
    optd={'MaxMainIter':it,'RelStopTol':0.0,'L':50.0}
    t0=time.time(); b = pc.ConvBPDN(D,S,0.05,pc.ConvBPDN.Options(optd)); X=b.solve(); t1=time.time()
    os.environ['SPORCO_AMD_OLD_ROWS']='1'
    b0 = pc.ConvBPDN(D,S,0.05,pc.ConvBPDN.Options(optd)); X0=b0.solve()
    os.environ.pop('SPORCO_AMD_OLD_ROWS')
    ref = orc.pgm_cbpdn(D.reshape(4,4,1,1,K), S.reshape(H,W,1,N,1), 0.05, dtype=np.float64, maxiter=it, L=50.0, rel_tol=0.0)
    print(H,W,K,N,'fused',b.dev.uses_fused_rows(), b0.dev.uses_fused_rows(),'time %.1f'%(t1-t0),'X vs generic',rel_l2(X,X0),'X vs ref',rel_l2(X,ref['X']))
    its=b.getitstat(); its0=b0.getitstat()
    for f in ('ObjFun','DFid','RegL1','Rsdl'): print('  ',f, rel_l2(getattr(its,f), ref[f]), rel_l2(getattr(its0,f), ref[f]))
    print('   Xf', rel_l2(b.Xf, np.fft.rfftn(ref['X'],axes=(0,1))), 'Yf vs generic', rel_l2(b.Yf, b0.Yf), 'Xfprv', rel_l2(b.Xfprv, b0.Xfprv), 'Yfprv', rel_l2(b.Yfprv, b0.Yfprv))
    b.solve(); b0.solve()
    print('   continue: X', rel_l2(b.X, b0.X))
