"""INTEGRATION.md section 2 shows the reference-side binding a maintainer would add
(`sporco/hip/__init__.py`: raw ctypes on include/sporco_amd.h).  This test EXECUTES that code block
verbatim -- extracted from the document, in a process that never imports the `sporco_amd` package --
and checks its result against a run of the unmodified reference (fixture admm_fixedrho_f64:
ConvBPDN, 25 iterations at fixed rho; sporco/admm/admm.py:293-389)."""

import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, build_hostsim

DRIVER = r'''
import sys, numpy as np
ns = {}
exec(compile(open(sys.argv[1]).read(), 'sporco/hip/__init__.py', 'exec'), ns)
assert 'sporco_amd' not in sys.modules
g = np.load(sys.argv[2])
rho = float(g['rho_final'])
opt = {'rho': rho, 'RelaxParam': 1.0, 'MaxMainIter': int(g['k_final'])}
assert ns['device_count']() >= 1
Y = ns['cbpdn'](g['D'], g['S'], float(g['lmbda']), opt)
ref = g['Y'].squeeze()
err = np.linalg.norm(Y - ref) / np.linalg.norm(ref)
print('STUB_REL_L2 %.3e' % err)
'''


def stub_source(libpath):
    doc = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    sec = doc[doc.index('## 2. A reference-side binding'):doc.index('## 3.')]
    code = re.search(r'```python\n(.*?)```', sec, flags=re.S).group(1)
    assert "ctypes.CDLL('libsporco_amd.so')" in code
    # the only edit: where the shared library is (a maintainer's install puts it on the loader path)
    return code.replace("ctypes.CDLL('libsporco_amd.so')", 'ctypes.CDLL(%r)' % libpath)


def run_stub(libpath, tmp_path):
    src = tmp_path / 'sporco_hip_init.py'
    src.write_text(stub_source(libpath))
    drv = tmp_path / 'driver.py'
    drv.write_text(DRIVER)
    r = subprocess.run([sys.executable, str(drv), str(src), os.path.join(GOLDEN, 'admm_fixedrho_f64.npz')],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    return float(re.search(r'STUB_REL_L2 (\S+)', r.stdout).group(1))


def test_integration_stub_on_simulator(tmp_path):
    assert run_stub(build_hostsim(), tmp_path) < 1e-4


@pytest.mark.gpu
def test_integration_stub_on_gpu(tmp_path):
    assert run_stub(os.path.join(REPO, 'sporco_amd', 'libsporco_amd.so'), tmp_path) < 1e-4
