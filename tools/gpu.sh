#!/bin/bash
# One parameterised launcher for every GPU session (replaces the per-session r0N_gpu_*.sh
# scripts of rounds 1-3, which are in the git history):
#
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh <tag> <step> [<step> ...]'
#
# Outputs go to gpurun_out/<tag>/ (copied by hand into profiles/<tag>_* when they are evidence).
# Steps (each bounded by its own `timeout`):
#   tests[:<pytest args>]  GPU test-suite (default: whole `-m gpu` suite)
#   smoke                  __graft_entry__.smoke()
#   bench[:<args>]         python bench.py <args>            -> bench[_<n>].json
#   quick[:<args>]         bench.py without CPU baseline / time-to-tol / parity / side configs
#   stats[:<args>]         rocprofv3 --kernel-trace --stats of the quick bench -> kernel_stats.csv
#   pmc                    FETCH_SIZE and WRITE_SIZE passes (separate runs) + hbm_traffic_bytes.json
#   sq                     two SQ counter passes of the quick bench -> pmc_sq_{1,2}.csv
#   py:<script and args>   python tools/<script> ...         -> <script>.jsonl (lines starting with {)
#   pystats:<script args>  rocprofv3 --kernel-trace --stats of python tools/<script> -> <script>_kernel_stats.csv
#   env:<VAR=VALUE>        export for the following steps (env:-VAR unsets)
#   sh:<command>           any command (bounded to 600 s)     -> sh_<n>.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
R=$PWD
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-time-to-tol --no-parity --configs none"
n=0
summ() { python $R/tools/rocpd_summary.py "$(find $1 -name '*.db' | head -1)" "$2" > /dev/null 2>&1 || echo "no rocpd summary for $2"; }
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  case $kind in
    env) if [ "${arg#-}" != "$arg" ]; then unset "${arg#-}"; else export "$arg"; fi ;;
    sh) timeout 600 bash -c "$arg" 2>&1 | grep -v amdgpu.ids | tee $O/sh_$n.txt | tail -40 ;;
    tests)
      timeout 2400 python -m pytest ${arg:-tests} -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -${PYTEST_TAIL:-6} | tee $O/pytest_gpu_$n.txt ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt ;;
    bench)
      timeout 1500 python bench.py $arg > $O/bench_$n.json 2> $O/bench_$n.err; tail -c 600 $O/bench_$n.err; head -c 700 $O/bench_$n.json; echo ;;
    quick)
      timeout 600 python bench.py $QUICK $arg 2>/dev/null | tail -1 > $O/quick_$n.json; head -c 500 $O/quick_$n.json; echo ;;
    stats)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks$n -o ks -- python $R/bench.py $QUICK $arg > $O/bench_under_rocprof_$n.json 2>/dev/null)
      summ /tmp/ks$n $O/kernel_stats_$n.csv; head -12 $O/kernel_stats_$n.csv | cut -c1-240 ;;
    pmc)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o ks -- python $R/bench.py $QUICK > /dev/null 2>&1)
      summ /tmp/pk $O/rocprofv3_kernel_stats.csv
      (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o f -- python $R/bench.py $QUICK --steps 6 --warmup 2 > /dev/null 2>&1)
      summ /tmp/pf $O/rocprofv3_pmc_fetch_size.csv
      (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o w -- python $R/bench.py $QUICK --steps 6 --warmup 2 > /dev/null 2>&1)
      summ /tmp/pw $O/rocprofv3_pmc_write_size.csv
      python tools/hbm_traffic_from_pmc.py $O/rocprofv3_pmc_fetch_size.csv $O/rocprofv3_pmc_write_size.csv $O/hbm_traffic_bytes.json "$TAG" $O/rocprofv3_kernel_stats.csv | tail -20 ;;
    sq)
      i=0
      for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"; do
        i=$((i+1))
        (cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace -d /tmp/sq$i -o c -- python $R/bench.py $QUICK --steps 6 --warmup 2 > /tmp/sq$i.log 2>&1)
        summ /tmp/sq$i $O/pmc_sq_$i.csv
      done ;;
    py)
      s=${arg%% *}
      timeout 900 python tools/$arg 2>&1 | grep "^{" | tee -a $O/${s%.py}.jsonl | cut -c1-400 ;;
    pystats)
      s=${arg%% *}
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/py$n -o ks -- python $R/tools/$arg > $O/${s%.py}_under_rocprof.txt 2>&1)
      summ /tmp/py$n $O/${s%.py}_kernel_stats.csv; head -12 $O/${s%.py}_kernel_stats.csv | cut -c1-240 ;;
    *) echo "unknown step $step" ;;
  esac
done
ls $O
