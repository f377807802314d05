# optimize() replay step after the round-3 changes (parallel bias-gradient finish, batched gelu' epilogue, context chain on a side stream):
# the backward tests, then the step timings with the side stream on / off
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03t
mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -8) > $OUT/pytest_train.log
cat $OUT/pytest_train.log
for cfg in "--batch 2 --size 1024 --train attn" "--batch 2 --size 1024 --train blocks" "--batch 8 --size 512 --train attn"; do
  timeout 600 python scripts/train_bench.py $cfg 2>/dev/null | tail -1 >> $OUT/train_bench.jsonl
done
MI355_TUNE="22=0" timeout 600 python scripts/train_bench.py --batch 2 --size 1024 --train attn 2>/dev/null | tail -1 > $OUT/train_bench_serial.json
cat $OUT/train_bench.jsonl $OUT/train_bench_serial.json
