#!/bin/bash
# Round 6: counters at the WEAK shapes -- the reference's 512^2 B = 2 CFG rollout and the SD3.5 optimize() step --
# one `rocprofv3 --pmc` pass per counter group (no trace domains beside it), summarised per kernel by scripts/summarize_prof.py:
#   MFMA-busy, LDS (instructions, bank conflicts, array-active cycles, LDS issue stalls vs wave cycles), FETCH_SIZE, WRITE_SIZE.
# usage: gpurun -- 'MID=<0|1> bash scripts/gpu_r6_pmc.sh'   (MID = mi355_tune_set key 32: the mid-size GEMM kernel off / on)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_ROUND=6
MID=${MID:-1}
LEAN="--no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --no-selfcheck --no-kernel-timing"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_WAVE_CYCLES\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_BUSY_CYCLES" | sort -u > gpurun_out/r06_pmc_counters_available.txt
for WL in small train; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r06pmc_${WL}_mid$MID; mkdir -p $OUT; rm -rf $OUT/prof_*
  if [ $WL = small ]; then CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --size 512 --batch 2 --guidance 4.5 --denoise-steps 2 $LEAN";
  else CMD="python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train default --iters 1 --only-step"; fi
  (cd /tmp && MI355_TUNE="8=0,22=0,32=$MID" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o run -- $CMD > $OUT/prof_stats.log 2>&1)
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $C | tr ' ' '_' | cut -c1-40)
    (cd /tmp && MI355_TUNE="8=0,22=0,32=$MID" timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- $CMD > $OUT/prof_pmc_$tag.log 2>&1); echo "pmc $WL $tag rc=$?"
  done
  python scripts/summarize_prof.py $OUT prof_stats > $OUT/summary.txt 2>&1
  find $OUT -type f -size +2M -delete
  head -n 60 $OUT/summary.txt | cut -c1-260
done
