#!/bin/bash
# FIRST GPU call of round 5: what had never run on a GPU (VERDICT r4 "next round" #1), before any new kernel.
#   (a) the Wan training step under the schedule race checker (MI355_RUN_UNVERIFIED=1), traces dumped for tests/golden/sched_traces/
#   (b) scripts/wan_train_bench.py at config D's own shape (480 x 832 x 49 frames = 20 280 tokens, CFG, full depth) + the 2-block gradient test at
#       that token count + the two ADVICE r4 cases (Nt_pad > S_pad, D = 5120)
#   (c) scripts/train_bench.py --train default (SD3_5Adapter.default_target_modules, sd3_5.py:75-80) + the full-width gradient test on that set
#   (d) the single-stream rocprof table with bench.py's key-8 restore fixed
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05a; mkdir -p $O
MI355_RUN_UNVERIFIED=1 MI355_DUMP_TRACES=$O/traces timeout 300 python -m pytest tests/test_gpu_schedules.py -q -s -m gpu -k "wan_training_step" > $O/pytest_wan_schedule.txt 2>&1; echo "rc=$?" >> $O/pytest_wan_schedule.txt
( time timeout 900 python -m pytest tests/test_gpu_wan_backward.py -q -s -m gpu --durations=8 ) > $O/pytest_wan_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_wan_backward.txt
timeout 600 python scripts/wan_train_bench.py --batch 1 --iters 2 > $O/wan_train_b1_480p49.json 2> $O/wan_train_b1_480p49.err; echo "rc=$?" >> $O/wan_train_b1_480p49.err
timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 3 > $O/train_default.json 2> $O/train_default.err; echo "rc=$?" >> $O/train_default.err
timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 3 > $O/train_attn.json 2>/dev/null
( time timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -s -m gpu -k "replay_gradients" ) > $O/pytest_fullsize_grads.txt 2>&1; echo "rc=$?" >> $O/pytest_fullsize_grads.txt
(cd /tmp && MI355_TUNE="8=0" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_single -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-families --no-train-step --no-vae > $O/prof_stats1.log 2>&1)
python scripts/summarize_prof.py $O prof_stats_single > $O/prof_summary_single_stream.txt 2>&1
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rc=\|no race\|races\|Error\|worst" $O/pytest_*.txt | cut -c1-400 | tail -n 30
tail -n 2 $O/*.json $O/*.err | cut -c1-900
head -n 30 $O/prof_summary_single_stream.txt
