"""Alternating dictionary learning driver (host side).

Same contract as ``sporco.dictlrn.dictlrn`` (IterStatsConfig:
sporco/dictlrn/dictlrn.py:28-160, DictLearn: :176-419): ``solve()`` alternates
``xstep.solve()`` and ``dstep.solve()`` (each usually one inner iteration),
coupling them through ``post_xstep`` / ``post_dstep``, and assembles an
``IterationStats`` row from the two inner solvers' latest rows.
"""

import collections

from .. import cdict
from .. import common
from .. import util

__all__ = ['IterStatsConfig', 'DictLearn']


class IterStatsConfig(object):
    """Which outer IterationStats field comes from where, plus display layout."""

    fwiter = 4
    fpothr = 2

    def __init__(self, isfld, isxmap, isdmap, evlmap, hdrtxt, hdrmap, fmtmap=None):
        self.IterationStats = collections.namedtuple('IterationStats', isfld)
        self.isxmap, self.isdmap, self.evlmap = isxmap, isdmap, evlmap
        self.hdrtxt, self.hdrmap = hdrtxt, hdrmap
        self.hdrstr, self.fmtstr, self.nsep = common.solve_status_str(
            hdrtxt, fmtmap=fmtmap, fwdth0=type(self).fwiter, fprec=type(self).fpothr)

    def iterstats(self, j, t, isx, isd, evl):
        row = []
        for name in self.IterationStats._fields:
            if name in self.isxmap:
                row.append(getattr(isx, self.isxmap[name]))
            elif name in self.isdmap:
                row.append(getattr(isd, self.isdmap[name]))
            elif name in self.evlmap:
                row.append(evl[name])
            elif name == 'Iter':
                row.append(j)
            elif name == 'Time':
                row.append(t)
            else:
                row.append(None)
        return self.IterationStats._make(row)

    def printheader(self):
        print(self.hdrstr)
        self.printseparator()

    def printseparator(self):
        print("-" * self.nsep)

    def printiterstats(self, itst):
        print(self.fmtstr % tuple(getattr(itst, self.hdrmap[col]) for col in self.hdrtxt))

    def __getstate__(self):
        d = self.__dict__.copy()
        d['_isfld'] = self.IterationStats._fields
        del d['IterationStats']
        return d

    def __setstate__(self, d):
        fields = d.pop('_isfld')
        self.__dict__.update(d)
        self.IterationStats = collections.namedtuple('IterationStats', fields)


class _DictLearnMeta(type):

    def __call__(cls, *args, **kwargs):
        obj = super(_DictLearnMeta, cls).__call__(*args, **kwargs)
        obj.timer.stop('init')
        return obj


class DictLearn(object, metaclass=_DictLearnMeta):
    """General alternation between a sparse coding and a dictionary update solver."""

    class Options(cdict.ConstrainedDict):
        defaults = {'Verbose': False, 'StatusHeader': True, 'IterTimer': 'solve',
                    'MaxMainIter': 1000, 'Callback': None}

        def __init__(self, opt=None):
            cdict.ConstrainedDict.__init__(self, {} if opt is None else opt)

    def __new__(cls, *args, **kwargs):
        obj = super(DictLearn, cls).__new__(cls)
        obj.timer = util.Timer(['init', 'solve', 'solve_wo_eval'])
        obj.timer.start('init')
        return obj

    def __init__(self, xstep, dstep, opt=None, isc=None):
        if opt is None:
            opt = DictLearn.Options()
        self.opt = opt
        if isc is None:
            isc = IterStatsConfig(
                isfld=['Iter', 'ObjFunX', 'XPrRsdl', 'XDlRsdl', 'XRho', 'ObjFunD',
                       'DPrRsdl', 'DDlRsdl', 'DRho', 'Time'],
                isxmap={'ObjFunX': 'ObjFun', 'XPrRsdl': 'PrimalRsdl',
                        'XDlRsdl': 'DualRsdl', 'XRho': 'Rho'},
                isdmap={'ObjFunD': 'DFid', 'DPrRsdl': 'PrimalRsdl',
                        'DDlRsdl': 'DualRsdl', 'DRho': 'Rho'},
                evlmap={},
                hdrtxt=['Itn', 'FncX', 'r_X', 's_X', u'ρ_X', 'FncD', 'r_D', 's_D', u'ρ_D'],
                hdrmap={'Itn': 'Iter', 'FncX': 'ObjFunX', 'r_X': 'XPrRsdl',
                        's_X': 'XDlRsdl', u'ρ_X': 'XRho', 'FncD': 'ObjFunD',
                        'r_D': 'DPrRsdl', 's_D': 'DDlRsdl', u'ρ_D': 'DRho'})
        self.isc = isc
        self.xstep = xstep
        self.dstep = dstep
        self.itstat = []
        self.j = 0

    @staticmethod
    def _latest(step):
        if step.itstat:
            return step.itstat[-1]
        return step.IterationStats(*([0.0] * len(step.IterationStats._fields)))

    def solve(self):
        """Alternate X and D updates (sporco/dictlrn/dictlrn.py:293-375)."""
        if self.opt['Verbose'] and self.opt['StatusHeader']:
            self.isc.printheader()
        self.timer.start(['solve', 'solve_wo_eval'])
        for self.j in range(self.j, self.j + self.opt['MaxMainIter']):
            self.xstep.solve()
            self.post_xstep()
            self.dstep.solve()
            self.post_dstep()
            self.timer.stop('solve_wo_eval')
            evl = self.evaluate()
            self.timer.start('solve_wo_eval')
            t = self.timer.elapsed(self.opt['IterTimer'])
            itst = self.isc.iterstats(self.j, t, self._latest(self.xstep),
                                      self._latest(self.dstep), evl)
            self.itstat.append(itst)
            if self.opt['Verbose']:
                self.isc.printiterstats(itst)
            if self.opt['Callback'] is not None:
                if self.opt['Callback'](self):
                    break
        self.j += 1
        self.timer.stop(['solve', 'solve_wo_eval'])
        if self.opt['Verbose'] and self.opt['StatusHeader']:
            self.isc.printseparator()
        return self.getdict()

    def post_xstep(self):
        self.dstep.setcoef(self.xstep.getcoef())

    def post_dstep(self):
        self.xstep.setdict(self.dstep.getdict())

    def evaluate(self):
        return None

    def getdict(self):
        return self.dstep.getdict()

    def getcoef(self):
        return self.xstep.getcoef()

    def getitstat(self):
        return util.transpose_ntpl_list(self.itstat)
