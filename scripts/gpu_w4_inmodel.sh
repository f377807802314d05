# in-model per-kernel durations, ping-pong only (0=3) vs default dispatch with the 4-wave kernel (0=1); two-stream off for clean attribution
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
for t in "0=3" "0=1"; do
  rm -rf $OUT/prof_stats
  (cd /tmp && MI355_TUNE="8=0,$t" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-vae > $OUT/prof_$t.log 2>&1)
  python scripts/summarize_prof.py $OUT > $OUT/summary_$t.txt 2>&1; head -16 $OUT/summary_$t.txt
  grep '^{' $OUT/prof_$t.log | cut -c1-200
done
rm -rf $OUT/prof_stats
