# optimize() replay with every block linear trainable: the context stream's weight gradients on the side stream (own scratch)
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03y
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x 2>&1 | tail -3) > $OUT/pytest.log
cat $OUT/pytest.log
timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train blocks 2>/dev/null | tail -1 > $OUT/train_blocks.json
MI355_TUNE="22=0" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train blocks 2>/dev/null | tail -1 > $OUT/train_blocks_serial.json
cat $OUT/train_blocks.json $OUT/train_blocks_serial.json
