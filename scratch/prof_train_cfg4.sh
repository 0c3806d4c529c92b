R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_train4
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload train --batch 16384 --ppo-steps 64 --ppo-epochs 4 --ppo-minibatches 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT/trace -name "*_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per iteration:', tot/3/1e6)
for r in rows[:28]:
    print('%-100s calls %5s avg_us %9.1f  pct %5.1f'%(r['Name'].replace('void ','').replace('at::native::','')[:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
