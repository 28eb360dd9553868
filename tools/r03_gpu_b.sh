#!/bin/bash
# Round 3, GPU call B: joint V form + W = 512 joint oracle tests, config 3 shard bench (V form vs
# (Y, U) form), then the whole GPU suite.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r03c
mkdir -p $O
timeout 900 python -m pytest tests/test_vform.py tests/test_parity_baseline_shapes.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/pytest_subset.txt
timeout 600 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids > $O/config3_vform.txt
SPORCO_AMD_NO_VFORM=1 timeout 600 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids | head -1 > $O/config3_yuform.txt
head -c 700 $O/config3_vform.txt; echo; tail -3 $O/config3_vform.txt; head -c 700 $O/config3_yuform.txt; echo
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/pytest_gpu.txt
