#!/bin/bash
# Round 3: SQ counter passes (instruction mix, wait / issue split, LDS) of the default bench command
# and of the FISTA bench, for the before / after comparison with profiles/r02c_rocprofv3_pmc_sq_*.csv.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/r03s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pm$i -o c -- $B --steps 6 --warmup 2 > /tmp/pm$i.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pm$i -name "*.db" | head -1) $O/pmc_sq_$i.csv > /dev/null 2>&1 || tail -5 /tmp/pm$i.log
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pg$i -o c -- python $R/tools/bench_other.py pgm > /tmp/pg$i.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pg$i -name "*.db" | head -1) $O/config4_pmc_sq_$i.csv > /dev/null 2>&1 || tail -5 /tmp/pg$i.log
done
ls -la $O
