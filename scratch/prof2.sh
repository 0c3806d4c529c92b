R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_rollout
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload rollout --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
echo rc=$?
f=$(find $OUT/trace -name "*_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time per step (us):', tot/45/1e3)
for r in rows[:22]:
    print('%-90s calls %5s avg_us %9.1f  pct %5.1f'%(r['Name'].replace('void ','')[:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
