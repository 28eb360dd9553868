cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03d; mkdir -p $O
run() { echo "== $1" >> $O/config3_knobs.txt; env $1 N=16 timeout 300 python tools/bench_config3.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-600 >> $O/config3_knobs.txt; }
run "X=0"
run "SPORCO_AMD_ROWS_PERSIST=2"
run "SPORCO_AMD_ROWS_PERSIST=2 SPORCO_AMD_ROWS_STAGGER_GROUPS=4 SPORCO_AMD_ROWS_STAGGER_SLEEPS=2"
run "SPORCO_AMD_ROWS_PERSIST=2 SPORCO_AMD_ROWS_STAGGER_GROUPS=8 SPORCO_AMD_ROWS_STAGGER_SLEEPS=4"
cat $O/config3_knobs.txt
B="python bench.py --no-cpu-baseline --no-time-to-tol --no-parity"
SPORCO_AMD_NO_SPECULATION=1 timeout 300 $B > $O/bench_nospec.json 2>/dev/null
SPORCO_AMD_NO_SPECULATION=1 SPORCO_AMD_ROWS_PERSIST=2 SPORCO_AMD_ROWS_STAGGER_GROUPS=4 SPORCO_AMD_ROWS_STAGGER_SLEEPS=2 timeout 300 $B > $O/bench_nospec_persist.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_nospec','bench_nospec_persist'):
    d=json.load(open('gpurun_out/r03d/%s.json'%f))
    print(f, round(d['value'],1), round(d['steady_state']['value'],1), {k:(v['avg_ms'],v['moved_frac']) for k,v in d['kernel_roofline'].items()})
PY
