#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_xstep.py -m gpu -x -q -k "slab or gradreg or many_filters or joint" 2>&1 | tail -5
echo coop; timeout 300 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids | head -3 | tee gpurun_out/r02r_config3_coop.json
echo two kernels; SPORCO_AMD_SLAB_COOP=0 timeout 300 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids | head -3
echo k128; timeout 300 python tools/bench_k128.py 2>&1 | grep -v amdgpu.ids | tail -4
echo k128 two kernels; SPORCO_AMD_SLAB_COOP=0 timeout 300 python tools/bench_k128.py 2>&1 | grep -v amdgpu.ids | tail -4
