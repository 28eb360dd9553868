#!/bin/bash
# Second part of the round-2 closing run: rocprofv3 summaries with the working-dispatch columns,
# PMC passes, bench under torch.distributed.run, the tests touched since r02_final.sh.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
timeout 900 python -m pytest tests/test_ccmodmd.py tests/test_fused_xstep.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest_gpu_part2.txt
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-time-to-tol --no-parity 2>/dev/null | tail -1 | cut -c1-400 | tee $O/bench_torchrun_1rank.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o ks -- $B > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/kernel_stats.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o f -- $B --steps 6 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p2 -name "*.db" | head -1) $O/pmc_fetch_size.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o w -- $B --steps 6 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p3 -name "*.db" | head -1) $O/pmc_write_size.csv > /dev/null 2>&1
cd $R
python tools/hbm_traffic_from_pmc.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/hbm_traffic_bytes.json "round 2 final" | tail -6
head -6 $O/kernel_stats.csv | cut -c1-60,150-400
