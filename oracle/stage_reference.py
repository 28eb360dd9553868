#!/usr/bin/env python3
"""Stage the UNMODIFIED reference package into the git-ignored oracle/_ref/.

TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by the product
(sporco_amd/), never committed (oracle/_ref/ is listed in .gitignore).

The reference is pure Python: there is nothing to compile, so "building the
checker" means copying the package's .py files where they lie under
/root/reference (read-only, authoring container only) into oracle/_ref/sporco/
so that they travel to the GPU box with the repo snapshot, like the built
libsporco_amd.so does.  There `bench.py`'s `cpu_baseline` leg runs
oracle/time_reference.py in a SUBPROCESS (PYTHONPATH = oracle/_stubs :
oracle/_ref) to time `sporco.admm.cbpdn.ConvBPDN.solve`
(sporco/admm/admm.py:293-389) on the GPU box's own host cores --
`cpu_baseline.kind = "reference"`.  When oracle/_ref is absent (a checkout
that never ran build() next to the reference) that leg is skipped and the
NumPy port of oracle/cbpdn_oracle.py is the only CPU baseline.

    python oracle/stage_reference.py          # called by __graft_entry__.build()

Copied: sporco/*.py and the sub-packages admm, pgm, dictlrn, prox (the hot
path's modules and what they import).  Not copied: data/ (images, 4 MB),
cupy/, cuda stubs, docs, examples, tests.
"""

import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('SPORCO_REFERENCE', '/root/reference')
DST = os.path.join(HERE, '_ref')
SUBPACKAGES = ('admm', 'pgm', 'dictlrn', 'prox')


def stage(verbose=True):
    src = os.path.join(REF, 'sporco')
    if not os.path.isdir(src):
        if verbose:
            print('stage_reference: %s not present (GPU box / plain checkout): nothing staged' % src)
        return False
    dst = os.path.join(DST, 'sporco')
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    os.makedirs(dst)
    n = 0
    for f in sorted(os.listdir(src)):
        if f.endswith('.py'):
            shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
            n += 1
    for sub in SUBPACKAGES:
        for root, dirs, files in os.walk(os.path.join(src, sub)):
            dirs[:] = [d for d in dirs if d != '__pycache__']
            rel = os.path.relpath(root, src)
            os.makedirs(os.path.join(dst, rel), exist_ok=True)
            for f in files:
                if f.endswith('.py'):
                    shutil.copyfile(os.path.join(root, f), os.path.join(dst, rel, f))
                    n += 1
    with open(os.path.join(DST, 'README'), 'w') as fh:
        fh.write('Unmodified copy of %s/sporco (%d .py files), staged by oracle/stage_reference.py.\n'
                 'Git-ignored; measurement infrastructure only (bench.py cpu_baseline).\n' % (REF, n))
    if verbose:
        print('stage_reference: %d files -> %s' % (n, dst))
    return True


if __name__ == '__main__':
    sys.exit(0 if stage() else 0)
