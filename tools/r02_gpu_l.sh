#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ccmod_ism_cg.py tests/test_ccmodmd.py tests/test_dictlearn.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/bench_dictlearn_dsteps.py 2>&1 | tee gpurun_out/r02l_dsteps.json
SPORCO_AMD_CG_HOST=1 timeout 300 python tools/bench_dictlearn_dsteps.py 2>&1 | grep "dmethod=cg" | tee gpurun_out/r02l_dsteps_cghost.json
