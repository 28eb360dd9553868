#!/usr/bin/env python3
"""Time the UNMODIFIED reference on this host's cores (bench.py `cpu_baseline`, kind "reference").

MEASUREMENT INFRASTRUCTURE ONLY.  Runs as a subprocess of bench.py's cpu_baseline leg (the
product process never imports the reference):

    python oracle/time_reference.py H W K "1,2" SECONDS

imports `sporco` from oracle/_ref (staged by oracle/stage_reference.py; falls back to
/root/reference in the authoring container) with oracle/_stubs standing in for the three
import-time dependencies this image lacks, and runs `sporco.admm.cbpdn.ConvBPDN(D, S, 0.05,
opt).solve()` (sporco/admm/admm.py:293-389) with default options and RelStopTol = 0 on N = 1, 2
of bench.make_problem's images, as many iterations as fit the time budget (at least 2).  pyFFTW is
not installed in this image, so the reference runs its own numpy.fft fallback
(sporco/fft.py:621-639) on one thread -- that is the reference's CPU path as it exists here.
Prints one JSON object: per run the iterations, `timer.elapsed('solve')` (what the reference
itself reports) and image-iterations/s.
"""

import json
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def main():
    H, W, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ns = [int(x) for x in sys.argv[4].split(',')]
    seconds = float(sys.argv[5])
    staged = os.path.join(HERE, '_ref')
    if os.path.isdir(os.path.join(staged, 'sporco')):
        src = staged
    elif os.path.isdir('/root/reference/sporco'):
        src = '/root/reference'
    else:
        print(json.dumps({'error': 'reference not staged (oracle/_ref absent)'}))
        return 0
    sys.path.insert(0, src)
    sys.path.insert(0, os.path.join(HERE, '_stubs'))
    sys.path.insert(0, REPO)
    warnings.filterwarnings('ignore')
    import numpy as np
    from sporco.admm import cbpdn
    import sporco
    import bench

    out = {'what': 'unmodified reference sporco.admm.cbpdn.ConvBPDN %s (numpy.fft fallback: pyFFTW '
                   'not installed; single thread), %dx%d K=%d float32, default options, '
                   "timer.elapsed('solve')" % (getattr(sporco, '__version__', '?'), H, W, K),
           'source': src, 'host_cores': os.cpu_count(), 'threads_used': 1, 'runs': []}
    budget = seconds / float(sum(ns))
    for n in ns:
        D, S = bench.make_problem(H, W, K, n, 0)
        # one iteration first: its cost sizes the run that is reported
        b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(
            {'MaxMainIter': 1, 'RelStopTol': 0.0, 'Verbose': False}))
        t0 = time.perf_counter()
        b.solve()
        t1 = time.perf_counter() - t0
        iters = max(2, min(50, int(budget * n / max(t1, 1e-6)) - 1))
        b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(
            {'MaxMainIter': iters, 'RelStopTol': 0.0, 'Verbose': False}))
        b.solve()
        t = b.timer.elapsed('solve')
        out['runs'].append({'images': n, 'iterations': iters, 'solve_seconds': t,
                            'seconds_per_iteration': t / iters,
                            'image_iterations_per_second': n * iters / t,
                            'Y_l2': float(np.linalg.norm(b.Y.astype(np.float64)))})
    print(json.dumps(out))
    return 0


if __name__ == '__main__':
    sys.exit(main())
