# PMC passes over the stand-alone recurrent kernel at 65 536 rows (scratch/bench_rc64.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_rc64
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scratch/bench_rc64.py > $OUT/trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/p_mfma -- python3 $R/scratch/bench_rc64.py > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p_lds -- python3 $R/scratch/bench_rc64.py > /dev/null 2>&1
find $OUT -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" ! -name "*.json" ! -name "*.log" -delete
grep -v amdgpu $OUT/trace.log | head -3
python3 - <<PY
import csv,glob,collections
for d in ("p_mfma","p_lds"):
    for f in glob.glob("$OUT/"+d+"/**/*_counter_collection.csv", recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rc64" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d,k,sum(v)/len(v), "n=",len(v))
PY
