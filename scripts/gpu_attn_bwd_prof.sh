# flash-attention backward inside the optimize() replay step: unit tests, then rocprofv3 per-kernel averages
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03w
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "attention_backward or ragged or side_stream" 2>&1 | tail -3) > $OUT/pytest.log
cat $OUT/pytest.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 2 --only-step > $OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $OUT/prof.log
python - $OUT/prof <<'PY' > $OUT/attn_bwd.txt
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats*.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):5.1f}%")
PY
cat $OUT/attn_bwd.txt
timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train attn 2>/dev/null | tail -1 > $OUT/train_bench.json; cat $OUT/train_bench.json
find $OUT -type f -size +1M -delete
