#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pgm -o pgm -- python $R/tools/bench_other.py pgm > /tmp/prof_pgm.log 2>&1
tail -2 /tmp/prof_pgm.log
python $R/tools/rocpd_summary.py $(find /tmp/prof_pgm -name "*.db" | head -1) $R/gpurun_out/r02p_config4_kernel_stats.csv
