"""Field maps tying the outer DictLearn statistics to the inner solvers'
(the role of sporco/dictlrn/common.py:18-133), as data tables."""

_EVL = {'ObjFun': 'ObjFun', 'DFid': 'DFid', 'RegL1': 'RegL1'}

_X = {  # xmethod -> (stats map, fields w/o and with backtracking, headers, header map)
    'admm': dict(map={'XPrRsdl': 'PrimalRsdl', 'XDlRsdl': 'DualRsdl', 'XRho': 'Rho'},
                 fld=(['XPrRsdl', 'XDlRsdl', 'XRho'],) * 2,
                 txt=(['r_X', 's_X', u'ρ_X'],) * 2,
                 hdr=({'r_X': 'XPrRsdl', 's_X': 'XDlRsdl', u'ρ_X': 'XRho'},) * 2),
    'pgm': dict(map={'X_F_Btrack': 'F_Btrack', 'X_Q_Btrack': 'Q_Btrack',
                     'X_ItBt': 'IterBTrack', 'X_L': 'L', 'X_Rsdl': 'Rsdl'},
                fld=(['X_L', 'X_Rsdl'],
                     ['X_F_Btrack', 'X_Q_Btrack', 'X_ItBt', 'X_L', 'X_Rsdl']),
                txt=(['L_X'], ['F_X', 'Q_X', 'It_X', 'L_X']),
                hdr=({'L_X': 'X_L'},
                     {'F_X': 'X_F_Btrack', 'Q_X': 'X_Q_Btrack', 'It_X': 'X_ItBt',
                      'L_X': 'X_L'})),
}
_D = {
    'pgm': dict(map={'Cnstr': 'Cnstr', 'D_F_Btrack': 'F_Btrack', 'D_Q_Btrack': 'Q_Btrack',
                     'D_ItBt': 'IterBTrack', 'D_L': 'L', 'D_Rsdl': 'Rsdl'},
                fld=(['D_L', 'D_Rsdl'],
                     ['D_F_Btrack', 'D_Q_Btrack', 'D_ItBt', 'D_L', 'D_Rsdl']),
                txt=(['L_D'], ['F_D', 'Q_D', 'It_D', 'L_D']),
                hdr=({'L_D': 'D_L'},
                     {'F_D': 'D_F_Btrack', 'Q_D': 'D_Q_Btrack', 'It_D': 'D_ItBt',
                      'L_D': 'D_L'})),
    'cns': dict(map={'Cnstr': 'Cnstr', 'DPrRsdl': 'PrimalRsdl', 'DDlRsdl': 'DualRsdl',
                     'DRho': 'Rho'},
                fld=(['DPrRsdl', 'DDlRsdl', 'DRho'],) * 2,
                txt=(['r_D', 's_D', u'ρ_D'],) * 2,
                hdr=({'r_D': 'DPrRsdl', 's_D': 'DDlRsdl', u'ρ_D': 'DRho'},) * 2),
}
_D['ism'] = _D['cg'] = _D['cns']      # the ADMM D-steps report the same fields


def _bt(opt, node, method):
    """Index 1 when the (pgm) inner solver uses backtracking, else 0."""
    return 1 if method == 'pgm' and opt[node, 'Backtrack'] is not None else 0


def evlmap(accdfid):
    return dict(_EVL) if accdfid else {}


def isxmap(xmethod, opt):
    m = dict(_X[xmethod]['map'])
    if not opt['AccurateDFid']:
        m.update(_EVL)
    return m


def isdmap(dmethod):
    return dict(_D[dmethod]['map'])


def isfld(xmethod, dmethod, opt):
    return (['Iter', 'ObjFun', 'DFid', 'RegL1', 'Cnstr'] +
            _X[xmethod]['fld'][_bt(opt, 'CBPDN', xmethod)] +
            _D[dmethod]['fld'][_bt(opt, 'CCMOD', dmethod)] + ['Time'])


def hdrtxt(xmethod, dmethod, opt):
    return (['Itn', 'Fnc', 'DFid', u'ℓ1', 'Cnstr'] +
            _X[xmethod]['txt'][_bt(opt, 'CBPDN', xmethod)] +
            _D[dmethod]['txt'][_bt(opt, 'CCMOD', dmethod)])


def hdrmap(xmethod, dmethod, opt):
    hdr = {'Itn': 'Iter', 'Fnc': 'ObjFun', 'DFid': 'DFid', u'ℓ1': 'RegL1', 'Cnstr': 'Cnstr'}
    hdr.update(_X[xmethod]['hdr'][_bt(opt, 'CBPDN', xmethod)])
    hdr.update(_D[dmethod]['hdr'][_bt(opt, 'CCMOD', dmethod)])
    return hdr
