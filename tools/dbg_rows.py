import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import numpy as np
from test_fused_xstep import problem, solve
from conftest import rel_l2, use_backend
use_backend(sys.argv[1] if len(sys.argv)>1 else 'hostsim')
from oracle import cbpdn_oracle as orc
for (H,W,K,N,it) in [(256,256,4,1,3),(256,512,6,1,3)]:
    D,S = problem(H,W,K,N,seed=H+K)
    optd={'MaxMainIter':it,'RelStopTol':0.0}
    t0=time.time(); b,Y = solve(D,S,optd); t1=time.time()
    b0,Y0 = solve(D,S,optd,unfused=True)
    ref = orc.admm_cbpdn(D.reshape(4,4,1,1,K), S.reshape(H,W,1,N,1), 0.05, dtype=np.float64, maxiter=it, rel_tol=0.0)
    print(H,W,K,N,'time %.1f'%(t1-t0),'Y vs unfused',rel_l2(Y,Y0),'Y vs ref',rel_l2(Y,ref['Y']),'X', rel_l2(b.X, ref['X']), 'U', rel_l2(b.U, ref['U']))
    its=b.getitstat(); 
    for f in ('ObjFun','DFid','RegL1','PrimalRsdl','DualRsdl','Rho'): print('  ',f, rel_l2(getattr(its,f), ref[f]))
