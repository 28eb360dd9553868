"""Host utilities: labelled wall-clock timers and small helpers.

``Timer`` / ``ContextTimer`` keep the interface of ``sporco.util.Timer``
(sporco/util.py:574-905) because every solver exposes ``obj.timer`` with the
labels ``init``, ``solve``, ``solve_wo_func``, ``solve_wo_rsdl`` (and
``solve_wo_btrack`` for PGM); the device calls made between start/stop are
synchronous, so the readings include GPU time.
"""

import collections
from timeit import default_timer as _now


def _as_list(labels):
    return list(labels) if isinstance(labels, (list, tuple)) else [labels]


class Timer(object):
    """A set of independent start/stop accumulators addressed by label."""

    def __init__(self, labels=None, dfltlbl='main', alllbl='all'):
        self.t0 = {}   # label -> start time or None when stopped
        self.td = {}   # label -> accumulated seconds
        self.dfltlbl = dfltlbl
        self.alllbl = alllbl
        if labels is not None:
            for lbl in _as_list(labels):
                self.td[lbl] = 0.0
                self.t0[lbl] = None

    def _select(self, labels):
        if labels is None:
            return [self.dfltlbl]
        if labels == self.alllbl:
            return list(self.t0.keys())
        return _as_list(labels)

    def start(self, labels=None):
        t = _now()
        for lbl in ([self.dfltlbl] if labels is None else _as_list(labels)):
            if lbl not in self.td:
                self.td[lbl] = 0.0
                self.t0[lbl] = None
            if self.t0[lbl] is None:
                self.t0[lbl] = t

    def stop(self, labels=None):
        t = _now()
        for lbl in self._select(labels):
            if lbl not in self.t0:
                raise KeyError('Unrecognized timer key %s' % lbl)
            if self.t0[lbl] is not None:
                self.td[lbl] += t - self.t0[lbl]
                self.t0[lbl] = None

    def reset(self, labels=None):
        for lbl in self._select(labels):
            if lbl not in self.t0:
                raise KeyError('Unrecognized timer key %s' % lbl)
            self.t0[lbl] = None
            self.td[lbl] = 0.0

    def elapsed(self, label=None, total=True):
        t = _now()
        if label is None:
            label = self.dfltlbl
            if label not in self.t0:
                return 0.0
        if label not in self.t0:
            raise KeyError('Unrecognized timer key %s' % label)
        te = 0.0 if self.t0[label] is None else t - self.t0[label]
        return te + self.td[label] if total else te

    def labels(self):
        return self.t0.keys()

    def __str__(self):
        t = _now()
        width = max([len(lbl) for lbl in self.t0] + [len(self.dfltlbl)]) + 2
        rows = ['%-*s  Accum.       Current' % (width, 'Label'), '-' * (width + 25)]
        for lbl in sorted(self.t0):
            cur = ' Stopped' if self.t0[lbl] is None else ' %.2e s' % (t - self.t0[lbl])
            rows.append('%-*s  %.2e s  %s' % (width, lbl, self.td[lbl], cur))
        return '\n'.join(rows) + '\n'


class ContextTimer(object):
    """``with ContextTimer(timer, label): ...`` (sporco/util.py:808-905)."""

    def __init__(self, timer=None, label=None, action='StartStop'):
        if action not in ('StartStop', 'StopStart'):
            raise ValueError('Unrecognized action %s' % action)
        self.timer = Timer() if timer is None else timer
        self.label = label
        self.action = action

    def __enter__(self):
        (self.timer.start if self.action == 'StartStop' else self.timer.stop)(self.label)
        return self

    def __exit__(self, typ, value, traceback):
        (self.timer.stop if self.action == 'StartStop' else self.timer.start)(self.label)
        return not typ

    def elapsed(self, total=True):
        return self.timer.elapsed(self.label, total=total)


def transpose_ntpl_list(lst):
    """List of namedtuples -> namedtuple of lists (sporco/array.py:209-231)."""
    if not lst:
        return None
    cls = collections.namedtuple(type(lst[0]).__name__, lst[0]._fields)
    return cls(*[[lst[k][l] for k in range(len(lst))] for l in range(len(lst[0]))])
