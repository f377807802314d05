# flash-attention backward: transposed-read kernels (key 23 = 1) vs the round-2 kernels (0): unit tests under both, then per-kernel averages
# inside the optimize() replay step (rocprofv3)
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "23=1" "23=0"; do
  tag=$(echo $cfg | tr -d '=,')
  (MI355_TUNE=$cfg timeout 300 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "attention_backward or ragged or side_stream" 2>&1 | tail -3) > $OUT/pytest_$tag.log
  cat $OUT/pytest_$tag.log
  cd /tmp
  MI355_TUNE=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 2 --only-step > $OUT/prof_$tag.log 2>&1
  cd $GRAFT_REPO_ROOT
  tail -1 $OUT/prof_$tag.log
  python - $OUT/prof_$tag $cfg <<'PY' >> $OUT/attn_bwd_tr_ab.txt
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats*.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("== MI355_TUNE=" + sys.argv[2])
for r in rows:
    if "attn_bwd" in r["Name"] or "attn_kernel" in r["Name"] or "transpose_kernel" in r["Name"]:
        print(f"{r['Name'][:80]:80s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
done
cat $OUT/attn_bwd_tr_ab.txt
find $OUT -type f -size +1M -delete
