Q="--no-cpu-baseline --no-time-to-tol --no-parity --configs config5"
for lib in "" sporco_amd/variants/libsporco_amd_contract.so sporco_amd/variants/libsporco_amd_r5.so "" sporco_amd/variants/libsporco_amd_contract.so; do
  SPORCO_AMD_LIBRARY=$lib python bench.py $Q 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); v=d['configs']['config5']
print('$lib' or 'r6', round(v['value'],1), {n:x['avg_ms'] for n,x in v['kernels'].items() if x['avg_ms']>0.2})"
done
