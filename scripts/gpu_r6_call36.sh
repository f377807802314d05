#!/bin/bash
# Round 6, call 36: counters for the pipelined attention-backward passes: MFMA-busy / LDS / wave cycles of the SD3.5 optimize() step (head_dim 64)
# and of the FLUX.1 step (head_dim 128), one rocprofv3 --pmc pass per counter group.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_ROUND=6
for WL in sd3 flux; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r06pmc2_$WL; mkdir -p $OUT; rm -rf $OUT/prof_*
  if [ $WL = sd3 ]; then CMD="python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train default --iters 1 --only-step"; TUNE="8=0,22=0";
  else CMD="python $GRAFT_REPO_ROOT/scripts/flux_train_bench.py --batch 1 --size 1024 --iters 1"; TUNE="26=0"; fi
  (cd /tmp && MI355_TUNE="$TUNE" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o run -- $CMD > $OUT/prof_stats.log 2>&1)
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
    tag=$(echo $C | tr ' ' '_' | cut -c1-40)
    (cd /tmp && MI355_TUNE="$TUNE" timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- $CMD > $OUT/prof_pmc_$tag.log 2>&1); echo "pmc $WL $tag rc=$?"
  done
  python scripts/summarize_prof.py $OUT prof_stats > $OUT/summary.txt 2>&1
  find $OUT -type f -size +2M -delete
  grep -n "attn" $OUT/summary.txt | cut -c1-260 | head -24
done
