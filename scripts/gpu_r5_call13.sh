#!/bin/bash
# Round 5, call 13: the per-kernel table of the 512^2 B = 2 CFG example shape AFTER the w4 minimum-grid rule (single stream), for DESIGN 14.6 / 14.11.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05m; mkdir -p $O
LEAN="--no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step"
(cd /tmp && MI355_TUNE="8=0" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_small -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10 --no-selfcheck --no-kernel-timing $LEAN > $O/prof_small.log 2>&1)
python scripts/summarize_prof.py $O prof_stats_small > $O/prof_summary_512_b2_cfg_single_stream_after_rule.txt 2>&1; head -n 14 $O/prof_summary_512_b2_cfg_single_stream_after_rule.txt
find $O -type f -size +1M -delete
