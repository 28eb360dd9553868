set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python bench.py --steps 100 > gpurun_out/final/bench_100.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o ks -- python $GRAFT_REPO_ROOT/bench.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final/kernel_stats.csv > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/p2 -name "*.db" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final/pmc_fetch_size.csv > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/p3 -name "*.db" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final/pmc_write_size.csv > /dev/null 2>&1
ls -la $GRAFT_REPO_ROOT/gpurun_out/final
