Q="--no-cpu-baseline --no-time-to-tol --no-parity"
for lib in "" sporco_amd/variants/libsporco_amd_r5.so "" sporco_amd/variants/libsporco_amd_r5.so; do
  SPORCO_AMD_LIBRARY=$lib python bench.py $Q --configs config3,config4,config5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
out={'lib':'$lib' or 'r6','value':round(d['value'],1),'steady':round(d['steady_state']['value'],1)}
for k,v in d['configs'].items():
    out[k]=round(v['value'],2); out[k+'_kern']={n:x['avg_ms'] for n,x in v['kernels'].items() if x['avg_ms']>0.2}
    if 'next_steps' in v: out[k+'_next']=round(v['next_steps']['value'],2)
print(json.dumps(out))"
done
