#!/usr/bin/env python3
"""Rho of the first iterations of the bench workload (config 2): which iterations change it --
what the speculative emission of the row spectra (csc_rows.h) bets on.  One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sporco_amd.admm import cbpdn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
D, S = bench.make_problem(512, 512, 64, 32, 0)
b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': n, 'RelStopTol': 0.0}))
b._return_min = False
b.solve()
st = b.getitstat()
rho = [float(x) for x in st.Rho]
r = [float(x) for x in st.PrimalRsdl]
s = [float(x) for x in st.DualRsdl]
print(json.dumps({'rho': rho, 'changed': [int(rho[i] != rho[i - 1]) for i in range(1, len(rho))],
                  'r_over_s': [r[i] / s[i] if s[i] else None for i in range(len(r))]}))
