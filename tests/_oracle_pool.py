"""Independent float64 oracle solves side by side on host cores -- test infrastructure (the oracle
is the checker, never the product): the full-size GPU parity tests compare every image of a
batch with an oracle run on that image alone, and those runs do not depend on each other."""

import multiprocessing as mp
from concurrent.futures import ProcessPoolExecutor
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve(job):
    """job = (positional arguments of oracle.cbpdn_oracle.admm_cbpdn, keyword arguments,
    {name: (field of the oracle's result, index into it, the device's array)}); returns
    ({name: relative l2 error of the device's array against the oracle's}, final ObjFun)."""
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from oracle import cbpdn_oracle as orc
    args, kw, checks = job
    ref = orc.admm_cbpdn(*args, **kw)
    errs = {}
    for name, (field, index, got) in checks.items():
        want = np.asarray(ref[field][index], dtype=np.float64)
        d = np.asarray(got, dtype=np.float64) - want
        errs[name] = float(np.sqrt(np.sum(d * d) / np.sum(want * want)))
    return errs, float(ref['ObjFun'][-1])


def oracle_solves(jobs, workers=16):
    """Run the jobs on min(workers, len(jobs), cores) processes ('spawn': the parent holds a HIP
    context, which does not survive a fork)."""
    n = max(1, min(workers, len(jobs), os.cpu_count() or 1))
    if n == 1:
        return [_solve(j) for j in jobs]
    # (an executor, not multiprocessing.Pool: a worker that dies makes it raise instead of
    # starting another one)
    with ProcessPoolExecutor(n, mp_context=mp.get_context('spawn')) as pool:
        return list(pool.map(_solve, jobs, timeout=1200))
