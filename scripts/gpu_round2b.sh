# Round-2 (second session) GPU pass, sized for a ~8-minute budget: (1) two-stream A/B in one process -> recommended MI355_TUNE,
# (2) the SD3.5 GPU tests (the engines touched this session) under that setting, (3) the headline bench under it, (4) if time is left,
# a rocprofv3 kernel-trace of the bench command and the remaining GPU tests.  Everything lands in gpurun_out/r2b/.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $OUT; rm -rf $OUT/*
date +%s > $OUT/t0
timeout 300 python scripts/two_stream_ab.py --quick --out $OUT > $OUT/ab.log 2>&1; echo "ab rc=$?" >> $OUT/status
tail -12 $OUT/ab.log
TUNE=$(cat $OUT/tune.env 2>/dev/null)
echo "TUNE=$TUNE" >> $OUT/status
export MI355_TUNE="$TUNE"
timeout 420 python -m pytest tests/test_gpu_rollout_variants.py tests/test_gpu_backward.py tests/test_gpu_model.py tests/test_gpu_adapter.py \
    tests/test_gpu_grpo_epoch.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -25 > $OUT/tests_sd3.log
echo "tests_sd3 rc=${PIPESTATUS[0]}" >> $OUT/status; tail -5 $OUT/tests_sd3.log
timeout 400 python bench.py 2>$OUT/bench.err > $OUT/bench.json; echo "bench rc=$?" >> $OUT/status; cut -c1-600 $OUT/bench.json
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck > $OUT/prof_stats.log 2>&1)
python scripts/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1; head -20 $OUT/prof_summary.txt
timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_rollout_variants.py --deselect tests/test_gpu_backward.py --deselect tests/test_gpu_model.py \
    --deselect tests/test_gpu_adapter.py --deselect tests/test_gpu_grpo_epoch.py --deselect tests/test_gpu_kernels.py --deselect tests/test_gpu_fullsize.py 2>&1 | tail -5 > $OUT/tests_rest.log
echo "tests_rest rc=${PIPESTATUS[0]}" >> $OUT/status; cat $OUT/tests_rest.log
find $OUT -type f -size +1M -delete
cat $OUT/status
