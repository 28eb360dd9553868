"""Multi-scale dictionaries -- a dictionary size `dsz` made of blocks of filters with supports of
their own, e.g. ((8, 8, 32), (12, 12, 32), (16, 16, 32)) (sporco/cnvrep.py:211-264, :729-817) --
in the host-side cnvrep functions, in the dictionary updates (the device's constraint projection
is told every filter's support: sporco_amd_csc_set_filter_sizes) and in dictionary learning
through the `DictSize` option, as the reference's examples/scripts/cdl/cbpdndl_pgm_clr.py does.
Fixtures: oracle/make_golden.py gen_multiscale (the unmodified reference)."""

import numpy as np
import pytest

from conftest import load_golden, rel_l2


def as_dsz(a):
    return tuple(tuple(int(v) for v in row) for row in a)


def test_cnvrep_functions():
    from sporco_amd import cnvrep as cr
    g = load_golden('cnvrep_multiscale')
    dsz = as_dsz(g['dsz'])
    ds = cr.DictionarySize(dsz)
    assert ds.nflt == 8 and ds.nchn == 1 and ds.mxsz == (8, 8)
    assert ds.fsz == [(4, 4)] * 3 + [(6, 6)] * 2 + [(8, 8)] * 3
    assert np.array_equal(cr.bcrop(g['v'], dsz), g['bcrop'])
    assert rel_l2(cr.zeromean(g['v'], dsz), g['zeromean']) < 1e-15
    assert rel_l2(cr.Pcn(g['v'], dsz, (16, 16), 2, 1, crp=False, zm=True), g['pcn']) < 1e-14
    assert rel_l2(cr.Pcn(g['v'], dsz, (16, 16), 2, 1, crp=True, zm=False), g['pcn_crop']) < 1e-14
    with pytest.raises(NotImplementedError):       # separate channel blocks: not taken
        cr.DictionarySize((((4, 4, 1, 2), (6, 6, 2, 2)),))


def test_pgm_dictionary_update(backend):
    from sporco_amd.pgm import ccmod
    g = load_golden('pgm_ccmod_multiscale_f64')
    c = ccmod.ConvCnstrMOD(g['Z'], g['S'], as_dsz(g['dsz']),
                           ccmod.ConvCnstrMOD.Options({'MaxMainIter': 15, 'ZeroMean': True, 'L': 50.0}))
    c.solve()
    D = c.getdict()
    assert D.shape == g['D'].shape and rel_l2(D, g['D']) < 1e-9
    # every filter is zero outside its own support
    assert np.all(D[4:, :, ..., 0:3] == 0) and np.all(D[6:, :, ..., 3:5] == 0)
    its = c.getitstat()
    for f in ('DFid', 'Rsdl'):
        assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f]) < 1e-9, f
    assert max(its.Cnstr) < 1e-12


def test_consensus_dictionary_update(backend):
    from sporco_amd.admm import ccmod
    g = load_golden('ccmod_cns_multiscale_f64')
    c = ccmod.ConvCnstrMOD_Consensus(g['Z'], g['S'], as_dsz(g['dsz']),
                                     ccmod.ConvCnstrMOD_Consensus.Options({'MaxMainIter': 12}))
    c.solve()
    assert rel_l2(c.getdict(), g['D']) < 1e-9 and rel_l2(c.Y, g['Y']) < 1e-9
    its = c.getitstat()
    for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), g['it_' + f]) < 1e-9, f


def test_dictionary_learning_multiscale_colour(backend):
    """ConvBPDNDictLearn (pgm / pgm) with DictSize = three scales of a colour dictionary."""
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden('cbpdndl_multiscale_clr_f64')
    opt = cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 8, 'DictSize': as_dsz(g['dsz']), 'CBPDN': {'L': 50.0}, 'CCMOD': {'L': 50.0}},
        xmethod='pgm', dmethod='pgm')
    d = cbpdndl.ConvBPDNDictLearn(g['D0'], g['S'], float(g['lmbda']), opt, xmethod='pgm', dmethod='pgm')
    D1 = d.solve()
    assert rel_l2(D1.squeeze(), g['D1'].squeeze()) < 1e-9
    assert rel_l2(d.getcoef(), g['X']) < 1e-9
    its = d.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1'):
        assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f]) < 1e-9, f
