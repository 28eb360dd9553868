#!/usr/bin/env python3
"""pgm.cbpdn.ConvBPDN at config 4's shape (512x512, K=64, N=32) under the step-size rules: fixed L,
BacktrackStandard, BacktrackRobust, StepSizePolicyCauchy / BB on the fused kernels, and the last
three composed from the staged calls (a subclass of the rule: the fused iteration does not restate a rule it does not
know).  One JSON line per variant."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sporco_amd.pgm import cbpdn as pc
from sporco_amd.pgm.backtrack import BacktrackStandard, BacktrackRobust


from sporco_amd.pgm.stepsize import StepSizePolicyCauchy, StepSizePolicyBB


class StagedRobust(BacktrackRobust):
    pass


class StagedCauchy(StepSizePolicyCauchy):
    pass


class StagedBB(StepSizePolicyBB):
    pass


N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
D, S = bench.make_problem(512, 512, 64, N, 0)
for name, bt in (('fixed L', None), ('BacktrackStandard (fused)', BacktrackStandard()),
                 ('BacktrackRobust (fused)', BacktrackRobust()),
                 ('BacktrackRobust (staged composition)', StagedRobust()),
                 ('StepSizePolicyCauchy (fused)', StepSizePolicyCauchy()),
                 ('StepSizePolicyCauchy (staged composition)', StagedCauchy()),
                 ('StepSizePolicyBB (fused)', StepSizePolicyBB()),
                 ('StepSizePolicyBB (staged composition)', StagedBB())):
    optd = {'MaxMainIter': 5, 'RelStopTol': 0.0, 'L': 500.0}
    if isinstance(bt, (StepSizePolicyCauchy, StepSizePolicyBB)):
        optd['StepSizePolicy'] = bt
        bt = None
    elif bt is not None:
        optd['Backtrack'] = bt
    b = pc.ConvBPDN(D, S, 0.05, pc.ConvBPDN.Options(optd))
    b._return_min = False
    b.solve(); b.dev.sync()
    b.opt['MaxMainIter'] = 20
    t0 = time.perf_counter(); b.solve(); b.dev.sync()
    dt = time.perf_counter() - t0
    st = b.getitstat()
    print(json.dumps({'config': 'pgm.cbpdn.ConvBPDN 512x512 K=64 N=%d f32, %s' % (N, name),
                      'fused': bool(b._fused_ok()), 'it_per_s': 20 / dt,
                      'trials_per_iteration': (float(np.mean(st.IterBTrack[-20:])) if bt is not None else None),
                      'L_final': float(b.L), 'ObjFun_final': float(st.ObjFun[-1])}))
    del b
