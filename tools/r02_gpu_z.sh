#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dictlearn.py tests/test_parity_baseline_shapes.py tests/test_onlinecdl.py tests/test_ccmod_cns.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/bench_dictlearn_k128.py 2>&1 | grep -v amdgpu.ids | grep "^{" | tee gpurun_out/r02z_dictlearn_k128.jsonl
