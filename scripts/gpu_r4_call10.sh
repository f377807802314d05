#!/bin/bash
# round 4, tenth GPU call: SD3.5 backward with the weight-gradient GEMMs on a third stream (key 26): bit identity, race check of the emitted
# three-stream schedule, the whole SD3 backward suite, and the A/B of the optimize() step (N1) at B = 2, 1024^2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16_grad_buffers.py -q -s -m gpu > $O/pytest_bf16_side.txt 2>&1; echo "rc=$?" >> $O/pytest_bf16_side.txt
MI355_DUMP_TRACES=$O/traces timeout 600 python -m pytest tests/test_gpu_schedules.py -q -s -m gpu -k "training_step" > $O/pytest_schedules.txt 2>&1; echo "rc=$?" >> $O/pytest_schedules.txt
timeout 900 python -m pytest tests/test_gpu_backward.py -q -m gpu > $O/pytest_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_backward.txt
for side in 1 0; do
  MI355_TUNE="26=$side" timeout 400 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 5 > $O/train_attn_side$side.json 2>/dev/null
  MI355_TUNE="26=$side" timeout 400 python scripts/train_bench.py --batch 2 --size 1024 --train blocks --iters 5 > $O/train_blocks_side$side.json 2>/dev/null
done
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rc=\|Error\|bit for bit\|no race\|races" $O/pytest_*.txt | cut -c1-300 | tail -n 20
tail -n 2 $O/*.json | cut -c1-900
