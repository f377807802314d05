#!/bin/bash
# FIRST GPU call of round 5 (written at the end of round 4, after its GPU budget was spent).  What has NOT run on a GPU yet:
#   * tests/test_gpu_schedules.py::test_wan_training_step_schedule_is_race_free  (the Wan training step's emitted launch list under the checker;
#     the launch_norm_rope_full forward kernel reports no regions of its own yet -- if the checker complains about unreported launches, add a
#     sched_trace_launch to launch_norm_rope_full in flux_ops.hip)
#   * scripts/wan_train_bench.py  (full-depth Wan2.1-1.3B optimize() step: timing, stash size, ratio_is_one)
#   * the whole -m gpu suite at a HEAD that includes the Wan native backward by default (the seven Wan backward tests were green in separate
#     calls: profiles/r04u_*, r04v_*; the rest of the suite at 1889a3d: profiles/r04_head_pytest_gpu_minus_fullsize.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05a; mkdir -p $O
MI355_RUN_UNVERIFIED=1 MI355_DUMP_TRACES=$O/traces timeout 300 python -m pytest tests/test_gpu_schedules.py -q -s -m gpu -k "wan_training_step" > $O/pytest_wan_schedule.txt 2>&1; echo "rc=$?" >> $O/pytest_wan_schedule.txt
timeout 600 python scripts/wan_train_bench.py --batch 1 --iters 2 > $O/wan_train_b1_480p49.json 2> $O/wan_train_b1_480p49.err; echo "rc=$?" >> $O/wan_train_b1_480p49.err
timeout 400 python scripts/wan_train_bench.py --batch 1 --frames 17 --iters 2 > $O/wan_train_b1_480p17.json 2>/dev/null
( time timeout 1800 python -m pytest tests -q -m gpu --durations=15 ) > $O/pytest_gpu_full.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_full.txt
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rc=\|no race\|races\|Error" $O/pytest_wan_schedule.txt $O/pytest_gpu_full.txt | cut -c1-300 | tail -n 12
tail -n 2 $O/*.json $O/*.err | cut -c1-900
