"""Placement of concurrently written arrays (sporco_amd/csrc/api_placement.inc) and the striped output
of the column pass (csc_fused.h FusedColsArgs::out_even / out_odd).  Where an array lies never
changes a result: with the search forced on at test sizes (SPORCO_AMD_PLACEMENT=force: it otherwise
serves arrays of 256 MiB and more) every solver must reproduce, bit for bit, what it computes with
the search off -- through the moved spectrum buffer, the chosen iterate buffers, the two half
spectra the column pass then writes, and the re-placed FISTA buffers."""

import os

import numpy as np
import pytest

from test_fused_xstep import problem


def run_admm(mode, host_loop, D, S, opt):
    from sporco_amd.admm import cbpdn
    os.environ['SPORCO_AMD_PLACEMENT'] = mode
    if host_loop:
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
    try:
        b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(opt))
        Y = b.solve()
        Y2 = b.solve()                  # (a second run: the single-array state carried over)
        rep = b._dev.placement_report()
        return Y, Y2, b.X.copy(), b.U.copy(), np.asarray(b.getitstat().ObjFun), np.asarray(b.getitstat().Rho), rep
    finally:
        os.environ.pop('SPORCO_AMD_PLACEMENT', None)
        os.environ.pop('SPORCO_AMD_HOST_LOOP', None)


@pytest.mark.parametrize('host_loop', [False, True])
def test_forced_placement_changes_no_bit_admm(backend, host_loop):
    H = 256 if backend == 'gpu' else 128
    D, S = problem(H, H, 8, 2 if backend == 'hostsim' else 3, seed=5)
    opt = {'MaxMainIter': 5, 'RelStopTol': 0.0}
    off = run_admm('0', host_loop, D, S, opt)
    on = run_admm('force', host_loop, D, S, opt)
    for a, c in zip(off[:6], on[:6]):
        assert np.array_equal(a, c)
    assert off[6] == []
    roles = [r['role'] for r in on[6]]
    assert 'V0' in roles and 'V1' in roles and 'cols_out_odd' in roles
    for r in on[6]:
        assert r['candidates'] >= 1 and r['bytes'] > 0 and r['chosen_ratio'] >= r['first_ratio'] - 1e-9


def test_forced_placement_changes_no_bit_pgm(backend):
    from sporco_amd.pgm import cbpdn as pc
    H = 256 if backend == 'gpu' else 128
    D, S = problem(H, H, 8, 2, seed=6)
    out = []
    for mode in ('0', 'force'):
        os.environ['SPORCO_AMD_PLACEMENT'] = mode
        try:
            b = pc.ConvBPDN(D, S, 0.05, pc.ConvBPDN.Options({'MaxMainIter': 5, 'RelStopTol': 0.0, 'L': 50.0}))
            X = b.solve()
            out.append((X, np.asarray(b.getitstat().ObjFun), b.dev.placement_report()))
        finally:
            os.environ.pop('SPORCO_AMD_PLACEMENT', None)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert out[0][2] == [] and any(r['role'].startswith('Yf') for r in out[1][2])
