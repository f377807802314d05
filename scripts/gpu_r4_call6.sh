#!/bin/bash
# round 4, sixth GPU call: Qwen-Image native backward (gradients vs oracle autograd, full-width blocks, optimizer step), FLUX.1 backward again after
# the move onto the shared helpers (train_common.h), and the optimize()-step timing of both at full depth
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_qwen_backward.py -x -q -s -m gpu > $O/pytest_qwen_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_qwen_backward.txt
timeout 900 python -m pytest tests/test_gpu_flux_backward.py -x -q -s -m gpu > $O/pytest_flux_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_flux_backward.txt
timeout 300 python -m pytest tests/test_gpu_qwen.py -x -q -m gpu > $O/pytest_qwen.txt 2>&1; echo "rc=$?" >> $O/pytest_qwen.txt
timeout 700 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 2 > $O/qwen_train_bench.json 2> $O/qwen_train_bench.err
timeout 400 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 2 > $O/flux_train_bench.json 2> $O/flux_train_bench.err
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rel-L2\|rc=\|Error\|error" $O/pytest_*.txt | cut -c1-400 | tail -n 40
tail -n 3 $O/*.json $O/*.err | cut -c1-1500
