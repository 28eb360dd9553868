"""The C-ABI boundary, checked without a GPU: the hipcc-built library (and the CPU simulator
build of the same sources) loads, exports every function include/sporco_amd.h declares, and the
Python binding knows each of them.  No compute calls."""

import ctypes
import os
import re
import subprocess

import pytest

from conftest import REPO, build_hostsim

HEADER = os.path.join(REPO, 'include', 'sporco_amd.h')
HIP_LIB = os.path.join(REPO, 'sporco_amd', 'libsporco_amd.so')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'^\s*(?:int|const char \*|void)\s*\*?\s*(sporco_amd_[a-z0-9_]+)\s*\(', src,
                       flags=re.M)
    assert len(names) > 60
    return sorted(set(names))


def test_header_and_binding_agree():
    from sporco_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_functions()


@pytest.mark.parametrize('which', ['hipcc', 'hostsim'])
def test_library_exports_every_declared_function(which):
    if which == 'hipcc':
        if not os.path.exists(HIP_LIB):
            subprocess.check_call(['make', '-s', '-C', os.path.join(REPO, 'sporco_amd', 'csrc'), '-j8'])
        path = HIP_LIB
    else:
        path = build_hostsim()
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:                      # e.g. no HIP runtime on this machine
        if which == 'hipcc':
            out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True,
                                 text=True, check=True).stdout
            exported = set(re.findall(r'\bT (sporco_amd_[a-z0-9_]+)', out))
            missing = [n for n in declared_functions() if n not in exported]
            assert not missing, missing
            pytest.skip("library not loadable here (%s); symbols checked with nm" % exc)
        raise
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    lib.sporco_amd_version.restype = ctypes.c_char_p
    assert re.search(rb'\d+\.\d+\.\d+', lib.sporco_amd_version())
    lib.sporco_amd_profile_slots.restype = ctypes.c_int
    assert lib.sporco_amd_profile_slots() > 0
