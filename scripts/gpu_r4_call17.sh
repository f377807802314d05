#!/bin/bash
# round 4: text chain of the FLUX.1 double blocks' backward on the plan's side stream (key 28): bit identity, gradients, race check, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04q; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_flux_backward.py tests/test_gpu_bf16_grad_buffers.py -q -m gpu > $O/pytest_flux_bf16.txt 2>&1; echo "rc=$?" >> $O/pytest_flux_bf16.txt
MI355_DUMP_TRACES=$O/traces timeout 300 python -m pytest tests/test_gpu_schedules.py -q -s -m gpu -k "head_dim_128" > $O/pytest_schedules.txt 2>&1; echo "rc=$?" >> $O/pytest_schedules.txt
for m in 1 0; do
  MI355_TUNE="28=$m" timeout 300 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 3 > $O/flux_text_side$m.json 2>/dev/null
  MI355_TUNE="28=$m" timeout 300 python scripts/flux_train_bench.py --batch 1 --size 512 --iters 3 > $O/flux512_text_side$m.json 2>/dev/null
done
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rc=\|Error\|no race\|races" $O/pytest_*.txt | cut -c1-300 | tail -n 10
for f in $O/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(d['ms_forward_backward'], d['ms_backward'], d['frac_of_2500'])")"; done
