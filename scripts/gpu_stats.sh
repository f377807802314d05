set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; rm -rf $OUT/prof_*
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1
head -30 $OUT/prof_summary.txt
find $OUT -type f -size +1M -delete
