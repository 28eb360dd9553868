#!/bin/bash
# Round-2 closing measurements on one MI355X: GPU test suite, smoke, bench lines (also under
# torch.distributed.run), rocprofv3 kernel statistics and the two PMC passes behind
# profiles/hbm_traffic_bytes.json, side benches.  Outputs: gpurun_out/final/ (copied to profiles/r02z_*).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 100 --no-cpu-baseline --no-time-to-tol --no-parity > $O/bench_100.json 2>/dev/null
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-time-to-tol --no-parity 2>/dev/null | tail -1 > $O/bench_torchrun_1rank.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o ks -- $B > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/kernel_stats.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o f -- $B --steps 6 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p2 -name "*.db" | head -1) $O/pmc_fetch_size.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o w -- $B --steps 6 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p3 -name "*.db" | head -1) $O/pmc_write_size.csv > /dev/null 2>&1
cd $R
python tools/hbm_traffic_from_pmc.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/hbm_traffic_bytes.json "round 2 final" > /dev/null 2>&1
timeout 300 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids | head -3 > $O/config3.json
timeout 300 python tools/bench_k128.py 2>&1 | grep -v amdgpu.ids | tail -2 > $O/k128.jsonl
timeout 300 python tools/bench_gradreg_k128.py 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/k128.jsonl
timeout 300 python tools/bench_pgm_k128.py 2>&1 | grep -v amdgpu.ids | grep "^{" >> $O/k128.jsonl
timeout 300 python tools/bench_other.py 2>&1 | grep -v amdgpu.ids | grep "^{" > $O/other.jsonl
timeout 300 python tools/bench_dictlearn_dsteps.py 2>&1 | grep -v amdgpu.ids | grep "^{" > $O/dsteps.jsonl
ls $O; head -c 300 $O/bench.json; echo; head -1 $O/config3.json | cut -c1-300; cat $O/other.jsonl
