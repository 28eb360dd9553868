#!/bin/bash
# round 2, call c: phase costs of fused_cols (copy-only / compute-only builds) and SQ counters
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
for d in 0 1 2; do
  SPORCO_AMD_COLS_DEBUG=$d timeout 300 $B --steps 40 --fastsolve > $O/bench_dbg$d.json 2>$O/err_dbg$d
  SPORCO_AMD_COLS_PERSIST=0 SPORCO_AMD_COLS_DEBUG=$d timeout 300 $B --steps 40 --fastsolve > $O/bench_np_dbg$d.json 2>>$O/err_dbg$d
done
python - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r02c/'
for n in ('bench_dbg0','bench_np_dbg0','bench_dbg1','bench_np_dbg1','bench_dbg2','bench_np_dbg2'):
    try:
        d=json.load(open(O+n+'.json')); print(n, round(d['value'],1), {k:v['avg_ms'] for k,v in d['kernel_roofline'].items()})
    except Exception as e: print(n, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o ks -- $B --steps 20 > /tmp/p1.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/kernel_stats.csv || tail -5 /tmp/p1.log
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pm$i -o c -- $B --steps 6 --warmup 2 > /tmp/pm$i.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pm$i -name "*.db" | head -1) $O/pmc_sq_$i.csv || tail -5 /tmp/pm$i.log
done
ls -la $O
