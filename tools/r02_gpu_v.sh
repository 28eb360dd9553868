#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_fused_xstep.py tests/test_parity_baseline_shapes.py -m gpu -x -q -k "joint or Joint or config3" 2>&1 | tail -3
timeout 300 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c90-330
timeout 300 python tools/bench_joint.py 2>&1 | grep -v amdgpu.ids | grep "^{" | cut -c1-200
