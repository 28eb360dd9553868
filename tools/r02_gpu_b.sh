#!/bin/bash
# round 2, call b: persistent fused_cols A/B, parity, SQ counters of the three kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02b; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
timeout 600 python -m pytest tests/test_parity_baseline_shapes.py tests/test_fused_xstep.py tests/test_ccmod_cns.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
SPORCO_AMD_COLS_PERSIST=0 timeout 300 $B > $O/bench_nopersist.json 2>$O/err1
timeout 300 $B > $O/bench_persist.json 2>$O/err2
SPORCO_AMD_COLS_PERSIST=0 timeout 300 $B --steps 100 > $O/bench_nopersist_100.json 2>>$O/err1
timeout 300 $B --steps 100 > $O/bench_persist_100.json 2>>$O/err2
python - <<'PY'
import json
for n in ('bench_nopersist','bench_persist','bench_nopersist_100','bench_persist_100'):
    try:
        d=json.load(open('gpurun_out/r02b/%s.json'%n)); print(n, round(d['value'],1), d['kernels_ms_per_iter'], {k:v['avg_ms'] for k,v in d['kernel_roofline'].items()})
    except Exception as e: print(n, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o ks -- $B --steps 20 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) $R/$O/kernel_stats.csv
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pm$i -o c -- $B --steps 6 --warmup 2 > /tmp/pm$i.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pm$i -name "*.db" | head -1) $R/$O/pmc_sq_$i.csv || tail -5 /tmp/pm$i.log
done
ls -la $R/$O
