"""Infrastructure shared by the iterative solver classes.

Follows the contract of ``sporco.common`` (sporco/common.py:88-294): a
metaclass builds the ``IterationStats`` namedtuple of every solver class from
its ``itstat_fields_*`` attributes and stops the ``init`` timer once
construction returns; ``set_dtype`` / ``set_attr`` give derived classes the
reference's "first writer wins unless reset" attribute semantics.
"""

import collections
import re

import numpy as np


class _SolverMeta(type):

    def __init__(cls, name, bases, ns):
        super(_SolverMeta, cls).__init__(name, bases, ns)
        stats = collections.namedtuple('IterationStats', cls.itstat_fields())
        # make the per-class namedtuple reachable for pickle
        stats.__module__ = cls.__module__
        stats.__qualname__ = cls.__qualname__ + '.IterationStats'
        cls.IterationStats = stats

    def __call__(cls, *args, **kwargs):
        obj = super(_SolverMeta, cls).__call__(*args, **kwargs)
        obj.timer.stop('init')
        return obj


class IterativeSolver(object, metaclass=_SolverMeta):
    """Base of all solver classes."""

    itstat_fields_objfn = ()
    itstat_fields_alg = ()
    itstat_fields_extra = ()

    @classmethod
    def itstat_fields(cls):
        return ('Iter',) + cls.itstat_fields_objfn + cls.itstat_fields_alg + \
            cls.itstat_fields_extra + ('Time',)

    def set_dtype(self, opt, dtype):
        """``opt['DataType']`` overrides ``dtype``; an existing value is kept."""
        if getattr(self, 'dtype', None) is None:
            self.dtype = np.dtype(dtype if opt['DataType'] is None else opt['DataType'])

    def set_attr(self, name, val, dval=None, dtype=None, reset=False):
        """Set ``self.<name>`` to ``val`` (or ``dval`` when ``val`` is None),
        converted to ``dtype``; without ``reset`` a non-None value is kept."""
        if val is None:
            val = dval
        if dtype is not None and val is not None:
            val = dtype(val) if isinstance(dtype, type) else dtype.type(val)
        if reset or getattr(self, name, None) is None:
            setattr(self, name, val)


def solve_status_str(hdrlbl, fmtmap=None, fwdth0=4, fwdthdlt=6, fprec=2):
    """Header, row format and width of the ``Verbose`` iteration table
    (sporco/common.py:230-294)."""
    fmtmap = fmtmap or {}
    fwdthn = fprec + fwdthdlt
    fmts = []
    for idx, lbl in enumerate(hdrlbl):
        if lbl in fmtmap:
            fmts.append(fmtmap[lbl])
        elif idx == 0:
            fmts.append('%%%dd' % fwdth0)
        else:
            fmts.append('%%%d.%de' % (fwdthn, fprec))
    widths = []
    for f in fmts:
        m = re.match(r'%-?(\d+)', f)
        if m is None:
            raise ValueError("Format string '%s' does not contain field width" % f)
        widths.append(int(m.group(1)))
    hdrstr = '  '.join('%-*s' % (w, t) for t, w in zip(hdrlbl, widths))
    return hdrstr, '  '.join(fmts), len(hdrstr)
