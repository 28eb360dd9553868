"""pytest configuration: markers, import paths, shared fixtures."""

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with -m gpu through gpurun)')
    _parallel_cpu_run(config)


def _parallel_cpu_run(config):
    """The CPU suite spends its time in the fiber simulator, one test at a time: for the
    plain `-m "not gpu"` run, spread the tests over a few worker processes (pytest-xdist,
    when installed).  GPU runs stay in one process.  SPORCO_AMD_TEST_WORKERS=0 disables it,
    an explicit -n wins."""
    if os.environ.get('PYTEST_XDIST_WORKER') or not config.pluginmanager.hasplugin('xdist'):
        return
    if (config.option.markexpr or '').strip() != 'not gpu':
        return
    if getattr(config.option, 'collectonly', False):
        return
    if getattr(config.option, 'numprocesses', None) is not None:
        build_hostsim()      # (an explicit -n: the workers must still not race for the build)
        return
    try:
        n = int(os.environ.get('SPORCO_AMD_TEST_WORKERS', min(8, os.cpu_count() or 1)))
    except ValueError:
        n = 0
    if n < 2:
        return
    build_hostsim()          # once, before the workers race for it
    config.option.numprocesses = n
    config.option.dist = 'load'
    config.option.tx = ['popen'] * n


def pytest_collection_modifyitems(config, items):
    """A case marked ``gpu`` through its parameters is meant for the real device only: drop
    its 'hostsim' twin (it would otherwise run the CPU simulator at GPU sizes under -m gpu)."""
    keep, drop = [], []
    for item in items:
        cs = getattr(item, 'callspec', None)
        if cs is not None and cs.params.get('backend') == 'hostsim' and \
                item.get_closest_marker('gpu') is not None:
            drop.append(item)
        else:
            keep.append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def rel_l2(a, b):
    """||a - b|| / ||b|| (relative l2 error against reference ``b``)."""
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b, dtype=a.dtype)
    nb = np.linalg.norm(b.ravel())
    if nb == 0.0:
        return float(np.linalg.norm(a.ravel()))
    return float(np.linalg.norm((a - b).ravel()) / nb)


# ---------------------------------------------------------------------------
# backends: the same parity tests run against
#   'hostsim' -- the kernel sources compiled for the CPU fiber simulator
#                (tests/hostsim; CPU-only runs, small sizes), and
#   'gpu'     -- the real hipcc/gfx950 library on an MI355X (-m gpu).
# ---------------------------------------------------------------------------
HOSTSIM_DIR = os.path.join(REPO, 'tests', 'hostsim')
HOSTSIM_LIB = os.path.join(HOSTSIM_DIR, 'libsporco_amd_hostsim.so')


def build_hostsim():
    # (SPORCO_AMD_HOSTSIM_LIB: a prebuilt simulator library to use instead -- the AddressSanitizer
    # build of tests/hostsim/Makefile's `asan` target)
    if os.environ.get('SPORCO_AMD_HOSTSIM_LIB'):
        return os.path.abspath(os.environ['SPORCO_AMD_HOSTSIM_LIB'])
    import subprocess
    subprocess.check_call(['make', '-s', '-C', HOSTSIM_DIR, '-j8'])
    return HOSTSIM_LIB


def use_backend(name):
    import sporco_amd
    from sporco_amd import _lib
    if name == 'hostsim':
        # (the simulator library carries a stand-in for librccl: single-rank, identity collectives)
        os.environ['SPORCO_AMD_RCCL_LIB'] = build_hostsim()
        sporco_amd.load_library(build_hostsim())
    else:
        if 'hostsim' in os.environ.get('SPORCO_AMD_RCCL_LIB', ''):
            os.environ.pop('SPORCO_AMD_RCCL_LIB')
        sporco_amd.load_library()      # in-tree libsporco_amd.so, hipcc build
        assert _lib.library_path().endswith('libsporco_amd.so')
        assert sporco_amd.device_count() > 0, "no AMD GPU visible"
        assert 'hostsim' not in sporco_amd.device_info(0)[0]
    return name


@pytest.fixture(params=['hostsim', pytest.param('gpu', marks=pytest.mark.gpu)])
def backend(request):
    return use_backend(request.param)


@pytest.fixture
def gpu_backend():
    return use_backend('gpu')
