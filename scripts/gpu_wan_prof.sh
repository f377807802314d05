set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ab
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_gpu_wan.py -m gpu -q -x -k "splits_static or measures_the_largest" 2>&1 | tail -3) > $OUT/pytest.log; cat $OUT/pytest.log
for cfg in "24=1" "24=0"; do
cd /tmp
MI355_TUNE=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$cfg -o t -- python $GRAFT_REPO_ROOT/scripts/wan_bench.py --batch 2 --denoise-steps 2 > $OUT/prof_$cfg.log 2>&1
cd $GRAFT_REPO_ROOT
python - $OUT/prof_$cfg $cfg <<'PY' >> $OUT/wan_kernels.txt
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats*.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("== MI355_TUNE=" + sys.argv[2])
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):5.1f}%")
PY
done
cat $OUT/wan_kernels.txt
find $OUT -type f -size +1M -delete
