#!/bin/bash
# Round 5, call 9: after the removal of the SD3.5 engine's third-stream weight-gradient opt-in (key 26 = 2): its backward tests, the schedule checks
# (traces re-recorded: two streams), the full-width gradient test, the optimize() step timing.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $O
( time MI355_DUMP_TRACES=$O/traces timeout 1200 python -m pytest tests/test_gpu_schedules.py tests/test_gpu_backward.py tests/test_gpu_bf16_grad_buffers.py tests/test_gpu_grpo_epoch.py tests/test_gpu_ddp_rccl.py -q -s -m gpu ) > $O/pytest_sd3_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_sd3_backward.txt
( time timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -s -m gpu -k "replay_gradients_vs_oracle" ) > $O/pytest_fullsize_grads.txt 2>&1; echo "rc=$?" >> $O/pytest_fullsize_grads.txt
timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 3 > $O/train_default.json 2>/dev/null
grep -h "passed\|failed\|rc=\|SD3.5 optimize\|Error" $O/pytest_sd3_backward.txt $O/pytest_fullsize_grads.txt | cut -c1-300 | tail -n 14
tail -n 1 $O/train_default.json | cut -c1-500; ls $O/traces | head -20
