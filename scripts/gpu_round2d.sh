# Round-2 (second session) closing evidence at the shipped defaults: rocprofv3 kernel trace of the bench command + the two-stream tests.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $OUT; rm -rf $OUT/*
export TMPDIR=/tmp
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-vae > $OUT/prof_stats.log 2>&1)
python scripts/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1; head -16 $OUT/prof_summary.txt
tail -1 $OUT/prof_stats.log | cut -c1-400
timeout 100 python -m pytest tests/test_gpu_rollout_variants.py -m gpu -q 2>&1 | tail -3 > $OUT/tests.log; cat $OUT/tests.log
find $OUT -type f -size +1M -delete
