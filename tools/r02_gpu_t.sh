#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_parity_baseline_shapes.py tests/test_dictlearn.py tests/test_onlinecdl.py tests/test_maskdcpl.py tests/test_device_loop.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/bench_other.py dl 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/bench_dictlearn_dsteps.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/bench_dictlearn_cns.py 2>&1 | grep -v amdgpu.ids | tail -3
