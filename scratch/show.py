import sys, json
for f in sys.argv[1:]:
    for l in open(f):
        try: d=json.loads(l)
        except Exception as e: print('bad line', f, l[:100]); continue
        print(d['config']['precision'], d['config']['launch'], d['config']['workload'][:36], '| value %.3e'%d['value'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'hbm %.4f'%d['roofline']['frac'], 'valu %.3f'%d['compute']['frac'], d.get('cpu_baseline',{}).get('value'))
