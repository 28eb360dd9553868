import json,sys
d=json.load(open(sys.argv[1]))
print('headline', round(d['value'],1), round(d['steady_state']['value'],1), d['roofline']['avg_kernel_ms'])
for k,c in (d.get('configs') or {}).items():
    if not isinstance(c,dict) or 'value' not in c: print(k, str(c)[:200]); continue
    print(k, round(c['value'],1), 'ms', round(c['ms_per_step'],3), 'par', c.get('parity',{}).get('pass'), 'dom', c['roofline'].get('kernel'), c['roofline'].get('avg_kernel_ms'), round(c['roofline'].get('frac',0),3), 'next', (c.get('next_steps') or {}).get('value'))
    print('   kernels', {kk:(round(v['avg_ms'],3)) for kk,v in c['kernels'].items()})
    if c.get('placement'): print('   placement', [(p['role'],p['candidates'],p['first_ratio'],p['chosen_ratio'],p['ms']) for p in c['placement']])
