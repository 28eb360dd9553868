#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_fused_xstep.py -m gpu -x -q -k "slab or gradreg or many_filters or joint" 2>&1 | tail -3
echo coop; timeout 300 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c90-330
echo k128; timeout 300 python tools/bench_k128.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/bench_gradreg_k128.py 2>&1 | grep -v amdgpu.ids | tail -2
