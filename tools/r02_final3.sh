#!/bin/bash
# Closing run, part 2: the full GPU suite on the final tree, the bench line, and rocprofv3
# evidence for the K > 64 column pass (config 3 shard): kernel statistics and HBM traffic.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/final3
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/pytest_gpu.txt
timeout 300 python bench.py > $O/bench.json 2>/dev/null
timeout 300 python tools/bench_dictlearn_dsteps.py 2>&1 | grep "^{" > $O/dsteps.jsonl
cd /tmp && export TMPDIR=/tmp
export N=8     # (8 of the 32 images of a shard: the capture stays small)
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/q1 -o ks -- python $R/tools/bench_config3.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/q1 -name "*.db" | head -1) $O/config3_n8_kernel_stats.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/q2 -o f -- python $R/tools/bench_config3.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/q2 -name "*.db" | head -1) $O/config3_n8_pmc_fetch_size.csv > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/q3 -o w -- python $R/tools/bench_config3.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/q3 -name "*.db" | head -1) $O/config3_n8_pmc_write_size.csv > /dev/null 2>&1
ls $O; head -c 250 $O/bench.json; echo; head -5 $O/config3_n8_kernel_stats.csv | cut -c1-90,200-330
grep "cols_slab_coop" $O/config3_n8_pmc_fetch_size.csv | cut -c1-60,230-400; grep "cols_slab_coop" $O/config3_n8_pmc_write_size.csv | cut -c1-60,230-400
