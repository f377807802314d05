#!/bin/bash
# round 4, ninth GPU call: weight-gradient GEMMs on a side stream (key 26): race check of the emitted FLUX.1 / Qwen-Image training schedules, bit
# identity vs the serial schedule, gradient tests again, and the A/B of the optimize() step at full depth
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16_grad_buffers.py -q -s -m gpu > $O/pytest_bf16_side.txt 2>&1; echo "rc=$?" >> $O/pytest_bf16_side.txt
MI355_DUMP_TRACES=$O/traces timeout 600 python -m pytest tests/test_gpu_schedules.py -q -s -m gpu -k "training_step" > $O/pytest_schedules.txt 2>&1; echo "rc=$?" >> $O/pytest_schedules.txt
timeout 900 python -m pytest tests/test_gpu_flux_backward.py tests/test_gpu_qwen_backward.py -q -m gpu > $O/pytest_backward128.txt 2>&1; echo "rc=$?" >> $O/pytest_backward128.txt
for side in 1 0; do
  MI355_TUNE="26=$side" timeout 400 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 3 > $O/flux_train_side$side.json 2>/dev/null
  MI355_TUNE="26=$side" timeout 700 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 3 > $O/qwen_train_side$side.json 2>/dev/null
done
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rc=\|Error\|bit for bit\|no race\|races" $O/pytest_*.txt | cut -c1-300 | tail -n 20
tail -n 2 $O/*.json | cut -c1-900
