import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import numpy as np
from test_fused_xstep import problem, solve
from conftest import rel_l2
import sporco_amd
sporco_amd.load_library()
for (H,W,K,N) in [(256,16,8,2),(512,12,64,1)]:
    D,S = problem(H,W,K,N,seed=H+K)
    optd={'MaxMainIter':1,'RelStopTol':0.0,'rho':2.0,'AutoRho':{'Enabled':False}}
    b,Y = solve(D,S,optd)
    b0,Y0 = solve(D,S,optd,unfused=True)
    print(H,W,K,N,'Y err',rel_l2(Y,Y0),'X err',rel_l2(b.X,b0.X), 'DFid', b.getitstat().DFid, b0.getitstat().DFid)
    X=b.X; X0=b0.X
    e = np.abs(X-X0).reshape(H,W,N,K)
    print(' err by k', e.max(axis=(0,1,2))[:8], ' by h%8', [float(e[i::8].max()) for i in range(8)])
