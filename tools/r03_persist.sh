#!/bin/bash
# One-launch solve of small problems (csc_rows.h admm_persist): tests and config 1 before / after.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03p
mkdir -p $O
timeout 900 python -m pytest tests/test_persist.py tests/test_device_loop.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/pytest.txt
for v in 0 1; do
  for i in 1 2; do SPORCO_AMD_PERSIST=$v timeout 200 python tools/bench_other.py c1 2>&1 | grep "^{" | sed "s/^{/{\"SPORCO_AMD_PERSIST\": $v, /" | tee -a $O/config1.jsonl; done
done
