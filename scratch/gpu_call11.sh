R=$GRAFT_REPO_ROOT
cd $R
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/final_smoke.log 2>&1; tail -n 5 gpurun_out/final_smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_form.json 2> gpurun_out/bench_driver_form.err ) 2> gpurun_out/bench_driver_time.log; cat gpurun_out/bench_driver_time.log
python -c "
import json
d=json.load(open('gpurun_out/bench_driver_form.json'))
print(d['value'], d['ms_per_step'], d['repeats'], d['roofline']['valu']['frac'], d['roofline']['frac'])
for k,v in d['extras'].items(): print(k, {kk:(round(vv,5) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','ms_per_step','ms_per_iteration','error')})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
