# Round-2 (second session), last short GPU pass (~4.5 min of budget): the shipped defaults (two-stream auto mode, fork after the block's
# last attention) -- A/B single / early fork / late fork on four shapes, the headline bench, and the tests that cover the changed paths.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p $OUT; rm -rf $OUT/*
AB_SHAPES="2x512x4.5x10x4,1x1024x1x28x2,2x1024x1x28x2,8x1024x1x28x2" timeout 200 python scripts/two_stream_ab.py --out $OUT > $OUT/ab.log 2>&1; echo "ab rc=$?" >> $OUT/status
tail -8 $OUT/ab.log
timeout 200 python bench.py --no-cpu-baseline --no-vae 2>$OUT/bench.err > $OUT/bench.json; echo "bench rc=$?" >> $OUT/status; cut -c1-300 $OUT/bench.json
timeout 300 python -m pytest tests/test_gpu_backward.py tests/test_gpu_rollout_variants.py tests/test_gpu_model.py tests/test_gpu_adapter.py -m gpu -q 2>&1 | tail -12 > $OUT/tests.log
echo "tests rc=${PIPESTATUS[0]}" >> $OUT/status; tail -4 $OUT/tests.log
cat $OUT/status
