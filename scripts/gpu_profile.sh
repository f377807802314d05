# GPU box: full GPU test suite + rocprofv3 kernel stats of the bench command + PMC passes.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
rm -rf $OUT/prof_* 
echo skip-tests > $OUT/t_gpu.log
cat $OUT/t_gpu.log
export TMPDIR=/tmp
cd /tmp
# 1. kernel trace + stats of the bench command (same command as the driver runs, fewer steps)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
tail -3 $OUT/prof_stats.log
find $OUT/prof_stats -name "*kernel_stats*" | head
# 2. PMC passes (own runs, short rollout: 2 denoise steps)
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-timing > $OUT/prof_pmc_$tag.log 2>&1
  tail -2 $OUT/prof_pmc_$tag.log
done
cd $GRAFT_REPO_ROOT
find $OUT -type f | head -60 > $OUT/prof_files.txt; cat $OUT/prof_files.txt
python scripts/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1
cat $OUT/prof_summary.txt
# keep only the small summaries in gpurun_out (the raw traces are large; the merge limit is 64 MiB)
find $OUT -type f -size +1M -delete
du -sh $OUT
