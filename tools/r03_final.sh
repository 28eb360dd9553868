#!/bin/bash
# Round-3 closing measurements on one MI355X: GPU test suite, smoke, bench lines (default, under
# torch.distributed.run with 1 RCCL rank, 2 gloo ranks sharing the GPU), rocprofv3 kernel statistics
# and the two PMC passes behind profiles/hbm_traffic_bytes.json, side benches.
# Outputs: gpurun_out/r03z/ (copied to profiles/r03z_*).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r03z
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
SPORCO_AMD_NO_VFORM=1 timeout 600 python bench.py --no-cpu-baseline --no-time-to-tol --no-parity > $O/bench_yuform.json 2>/dev/null
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-time-to-tol --no-parity 2>/dev/null | tail -1 > $O/bench_torchrun_1rank.json
SPORCO_AMD_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-time-to-tol --no-parity 2>/dev/null | tail -1 > $O/bench_gloo_2ranks_sharing_1gpu_correctness_only.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o ks -- $B > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/rocprofv3_kernel_stats.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o f -- $B --steps 6 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p2 -name "*.db" | head -1) $O/rocprofv3_pmc_fetch_size.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o w -- $B --steps 6 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p3 -name "*.db" | head -1) $O/rocprofv3_pmc_write_size.csv > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o ks -- python $R/tools/bench_other.py pgm > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p4 -name "*.db" | head -1) $O/config4_rocprofv3_kernel_stats.csv > /dev/null 2>&1
cd $R
python tools/hbm_traffic_from_pmc.py $O/rocprofv3_pmc_fetch_size.csv $O/rocprofv3_pmc_write_size.csv $O/hbm_traffic_bytes.json "round 3 final" $O/rocprofv3_kernel_stats.csv > /dev/null 2>&1
timeout 600 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids > $O/config3.txt
timeout 300 python tools/bench_other.py 2>&1 | grep "^{" > $O/other_configs.jsonl
timeout 300 python tools/bench_k128.py 2>&1 | grep -v amdgpu.ids | tail -2 > $O/k128.jsonl
timeout 300 python tools/bench_gradreg_k128.py 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/k128.jsonl
timeout 300 python tools/bench_pgm_k128.py 2>&1 | grep "^{" >> $O/k128.jsonl
for t in bench_ams bench_gradreg bench_joint bench_mcdict bench_grdmsk bench_size128; do timeout 300 python tools/$t.py 2>/dev/null | grep "^{" >> $O/side_benches.jsonl; done
timeout 300 python tools/bench_masked.py 2>/dev/null | grep "^{" > $O/masked.jsonl
timeout 300 python tools/bench_dictlearn_dsteps.py 2>&1 | grep "^{" > $O/dictlearn_dsteps.jsonl
timeout 300 python tools/bench_dictlearn_colour.py 2>&1 | grep "^{" > $O/dictlearn_colour.jsonl
# small problems as one launch (opt-in) against the launch-per-pass loop; the generic chain
for v in 0 1; do SPORCO_AMD_PERSIST=$v timeout 200 python tools/bench_other.py c1 2>&1 | grep "^{" | sed "s/^{/{\"SPORCO_AMD_PERSIST\": $v, /" >> $O/config1_one_launch.jsonl; done
timeout 400 python tools/bench_generic.py 2>&1 | grep "^{" > $O/generic_chain.jsonl
ls $O; head -c 400 $O/bench.json; echo; head -1 $O/config3.txt | cut -c1-300; cat $O/other_configs.jsonl
