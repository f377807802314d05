# Round-2 (second session), last GPU seconds: three-stream variant (key 11) -- bitwise test, then A/B single / two-stream / three-stream.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2e
mkdir -p $OUT; rm -rf $OUT/*
timeout 60 python -m pytest tests/test_gpu_rollout_variants.py -m gpu -q -k two_stream 2>&1 | tail -6 > $OUT/tests.log; cat $OUT/tests.log
AB_MODES="0,1,3" AB_SHAPES="2x512x4.5x10x4,8x512x1x10x4,1x1024x1x28x2,2x1024x1x28x2,4x1024x1x28x2" timeout 110 python scripts/two_stream_ab.py --out $OUT > $OUT/ab.log 2>&1; echo "ab rc=$?" >> $OUT/status
tail -9 $OUT/ab.log
