#!/bin/bash
# Round 3: rocprofv3 kernel statistics and the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the
# default bench command, summarised into profiles-ready files under gpurun_out/r03p/.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r03p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o ks -- $B > $O/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/kernel_stats.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o f -- $B --steps 6 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p2 -name "*.db" | head -1) $O/pmc_fetch_size.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o w -- $B --steps 6 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p3 -name "*.db" | head -1) $O/pmc_write_size.csv > /dev/null 2>&1
cd $R
python tools/hbm_traffic_from_pmc.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/hbm_traffic_bytes.json "round 3" $O/kernel_stats.csv
head -8 $O/kernel_stats.csv | cut -c1-260
