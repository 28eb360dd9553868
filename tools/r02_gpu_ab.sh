#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_xstep.py tests/test_device_loop.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/bench_size128.py 2>&1 | grep -v amdgpu.ids | grep "^{" | tee gpurun_out/r02ab_size128.jsonl
