#!/usr/bin/env python3
"""Time the UNMODIFIED reference (sporco.admm.cbpdn.ConvBPDN, numpy.fft fallback, single thread) on
the bench workload's images in the authoring container: 512x512, K=64, N = 1, 2, 4 of
bench.make_problem's images, default options, RelStopTol = 0.  Writes
profiles/r02_reference_cpu.json, which bench.py quotes beside its cpu_baseline ("reference_here").
The reference does not exist on the GPU box, so this cannot run there."""
import json
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.environ.get('SPORCO_REFERENCE', '/root/reference'))
sys.path.insert(0, os.path.join(REPO, 'oracle', '_stubs'))
sys.path.insert(0, REPO)
warnings.filterwarnings('ignore')
from sporco.admm import cbpdn   # noqa: E402
import bench                    # noqa: E402

out = {'what': 'unmodified reference sporco.admm.cbpdn.ConvBPDN (numpy.fft fallback, 1 thread), '
               '512x512 K=64 float32, default options, timer.elapsed("solve")',
       'host': 'authoring container: %d vCPUs (%s)' % (os.cpu_count(), os.uname().machine),
       'runs': []}
for n, iters in ((1, 6), (2, 5), (4, 4)):
    D, S = bench.make_problem(512, 512, 64, n, 0)
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': iters, 'RelStopTol': 0.0,
                                                           'Verbose': False}))
    t0 = time.perf_counter()
    b.solve()
    wall = time.perf_counter() - t0
    t = b.timer.elapsed('solve')
    out['runs'].append({'images': n, 'iterations': iters, 'solve_seconds': t, 'wall_seconds': wall,
                        'seconds_per_iteration': t / iters,
                        'image_iterations_per_second': n * iters / t})
    print(out['runs'][-1], flush=True)
r = out['runs'][-1]
out['iterations_per_second_at_N32_extrapolated'] = r['image_iterations_per_second'] / 32.0
with open(os.path.join(REPO, 'profiles', 'r02_reference_cpu.json'), 'w') as f:
    json.dump(out, f, indent=1)
