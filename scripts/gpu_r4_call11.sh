#!/bin/bash
# round 4, eleventh GPU call: modelled split-K factor of the weight-gradient GEMMs (key 27) A/B on the three families' optimize() steps; the
# training-schedule race checks and the gradient suites on the new defaults
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16_grad_buffers.py -q -s -m gpu > $O/pytest_bf16_side.txt 2>&1; echo "rc=$?" >> $O/pytest_bf16_side.txt
MI355_DUMP_TRACES=$O/traces timeout 600 python -m pytest tests/test_gpu_schedules.py -q -s -m gpu -k "training_step" > $O/pytest_schedules.txt 2>&1; echo "rc=$?" >> $O/pytest_schedules.txt
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_flux_backward.py tests/test_gpu_qwen_backward.py -q -m gpu > $O/pytest_backward_all.txt 2>&1; echo "rc=$?" >> $O/pytest_backward_all.txt
for m in 1 0; do
  MI355_TUNE="27=$m" timeout 400 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 5 > $O/sd3_attn_split$m.json 2>/dev/null
  MI355_TUNE="27=$m" timeout 400 python scripts/train_bench.py --batch 2 --size 1024 --train blocks --iters 5 > $O/sd3_blocks_split$m.json 2>/dev/null
  MI355_TUNE="27=$m" timeout 400 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 3 > $O/flux_split$m.json 2>/dev/null
  MI355_TUNE="27=$m" timeout 700 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 3 > $O/qwen_split$m.json 2>/dev/null
done
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rc=\|Error\|bit for bit\|no race\|races" $O/pytest_*.txt | cut -c1-300 | tail -n 20
for f in $O/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(d['ms_forward_backward'], d['ms_backward'], d['frac_of_2500'])")"; done
