# N1 evidence: full-size gradient parity, training-step timings, rocprofv3 kernel stats of one optimize() replay step
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k gradients 2>&1 | tail -15) > $OUT/pytest_fullgrad.log
for cfg in "--batch 2 --size 1024 --train attn" "--batch 2 --size 1024 --train blocks" "--batch 8 --size 512 --train attn" "--batch 4 --size 512 --train attn --guidance 4.5"; do
  timeout 600 python scripts/train_bench.py $cfg 2>/dev/null | tail -1 >> $OUT/train_bench.jsonl
done
cat $OUT/train_bench.jsonl
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train blocks --iters 2 > $OUT/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > $OUT/train_kernel_stats.txt 2>&1
import csv, glob, os, sys
sys.path.insert(0, "scripts")
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r02c")
f = glob.glob(os.path.join(out, "prof_train", "**", "*kernel_stats*.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("== rocprofv3 --kernel-trace --stats: scripts/train_bench.py --batch 2 --size 1024 --train blocks --iters 2")
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[:40]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
PY
head -45 $OUT/train_kernel_stats.txt
find $OUT -type f -size +1M -delete
