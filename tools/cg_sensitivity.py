"""Sensitivity of masked dictionary learning (xmethod admm, dmethod cg, default CG StopTol) to the
summation order of the CG operator: the reference against itself with linalg.inner summing the
filter axis in reversed order, and against sporco_amd."""
import sys, numpy as np, warnings
warnings.filterwarnings('ignore')
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/reference'); sys.path.insert(0,'/root/repo/oracle/_stubs')
import conftest, sporco_amd
sporco_amd.load_library(conftest.build_hostsim())
import sporco.linalg as rl
from sporco.dictlrn import cbpdndlmd as rdl
from sporco_amd.dictlrn import cbpdndlmd as odl
seed=int(sys.argv[1]); its=int(sys.argv[2])
rng=np.random.RandomState(seed)
N=4; S=rng.randn(16,16,N); W=(rng.rand(16,16,N)>0.2).astype(float); D0=rng.randn(5,5,6)
_inner = rl.inner
def inner_rev(x, y, axis=-1):
    xr = np.flip(x, axis=axis) if x.shape[axis] > 1 else x
    yr = np.flip(y, axis=axis) if y.shape[axis] > 1 else y
    return _inner(xr, yr, axis=axis)
def run(mod, patch=False):
    Ds=[]
    def cbk(d): Ds.append(d.getdict().copy()); return False
    if patch: rl.inner = inner_rev
    try:
        opt=mod.ConvBPDNMaskDictLearn.Options({'MaxMainIter':its,'Callback':cbk}, xmethod='admm', dmethod='cg')
        d=mod.ConvBPDNMaskDictLearn(D0,S,0.1,W,opt,xmethod='admm',dmethod='cg')
        d.solve()
    finally:
        rl.inner = _inner
    return Ds
ref=run(rdl); refp=run(rdl, True); our=run(odl)
rel=lambda a,b: np.linalg.norm(a.squeeze()-b.squeeze())/np.linalg.norm(a)
for i in range(its):
    print(i, 'ref vs ref(reversed sum) %.2e   ref vs sporco_amd %.2e' % (rel(ref[i],refp[i]), rel(ref[i],our[i])))
