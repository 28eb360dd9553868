#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_device_loop.py tests/test_parity_baseline_shapes.py tests/test_dist_nccl.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1; do echo "merge ctl = $v"; SPORCO_AMD_RUN_MERGE_CTL=$v timeout 300 python tools/bench_other.py c1 2>&1 | grep -v amdgpu.ids; done
timeout 300 python bench.py --no-cpu-baseline --no-time-to-tol --no-parity 2>/dev/null | cut -c1-200
