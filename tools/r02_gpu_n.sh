#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
SPORCO_AMD_CG_TICKETS=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_cg -o cg -- python $R/tools/bench_cg_only.py > /tmp/prof_cg.log 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/prof_cg -name "*.db" | head -1) rows_fwd 12 > $R/gpurun_out/r02n_cg_timeline.txt
