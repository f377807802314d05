#!/bin/bash
# round 4, fourteenth GPU call: FLUX.1 training forward with the text chain on the plan's side stream (bit identity vs the no-grad forward is asserted
# by the gradient tests), the key-26 / key-28 interaction fix, the race checks, one step timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flux_backward.py tests/test_gpu_bf16_grad_buffers.py -q -m gpu > $O/pytest_flux_bf16.txt 2>&1; echo "rc=$?" >> $O/pytest_flux_bf16.txt
MI355_DUMP_TRACES=$O/traces timeout 600 python -m pytest tests/test_gpu_schedules.py -q -s -m gpu -k "head_dim_128 or flux_two_stream" > $O/pytest_schedules.txt 2>&1; echo "rc=$?" >> $O/pytest_schedules.txt
timeout 400 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 3 > $O/flux_train.json 2>/dev/null
timeout 400 python scripts/flux_train_bench.py --batch 1 --size 512 --iters 3 > $O/flux_train_512.json 2>/dev/null
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rc=\|Error\|no race\|races" $O/pytest_*.txt | cut -c1-300 | tail -n 12
tail -n 1 $O/flux_train.json | cut -c1-700; tail -n 1 $O/flux_train_512.json | cut -c1-700
