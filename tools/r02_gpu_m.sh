#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ccmod_ism_cg.py tests/test_ccmodmd.py tests/test_dictlearn.py -m gpu -x -q 2>&1 | tail -5
echo tickets; timeout 300 python tools/bench_cg_only.py 2>&1 | grep -v amdgpu.ids
echo separate ctl; SPORCO_AMD_CG_TICKETS=0 timeout 300 python tools/bench_cg_only.py 2>&1 | grep -v amdgpu.ids
echo host; SPORCO_AMD_CG_HOST=1 timeout 300 python tools/bench_cg_only.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cg -o cg -- python $R/tools/bench_cg_only.py > /tmp/prof_cg.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_cg -name "*.db" | head -1) $R/gpurun_out/r02m_cg_kernel_stats.csv
SPORCO_AMD_CG_TICKETS=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cg2 -o cg -- python $R/tools/bench_cg_only.py > /tmp/prof_cg.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_cg2 -name "*.db" | head -1) $R/gpurun_out/r02m_cg_kernel_stats_sepctl.csv
