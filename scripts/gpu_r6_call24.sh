#!/bin/bash
# Round 6, call 24: row-major weight-gradient GEMMs for ragged M and for the head_dim-128 engines (train_common.h): operator test incl. ragged M,
# the backward suites of all four families, and the optimize() steps under key 39 = 1 (row-major) / 0 (transposed copies).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06w; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "row_major" ) > $O/pytest_wgrad.txt 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest_wgrad.txt
grep -h "passed\|failed\|rc=\|FAILED\|Error\|assert" $O/pytest_wgrad.txt | cut -c1-300 | tail -n 8
if [ $rc -ne 0 ]; then exit 0; fi
( time timeout 1800 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16_grad_buffers.py tests/test_gpu_flux_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_wan_backward.py -x -q -m gpu ) > $O/pytest_backward_all.txt 2>&1; echo "rc=$?" >> $O/pytest_backward_all.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error" $O/pytest_backward_all.txt | cut -c1-300 | tail -n 8
for t in "39=1" "39=0" "39=1" "39=0"; do
  MI355_TUNE="$t" timeout 300 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('flux1 tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_ab.txt
  MI355_TUNE="$t" timeout 300 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('qwen tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_ab.txt
  MI355_TUNE="$t" timeout 300 python scripts/wan_train_bench.py --batch 1 --frames 49 --iters 2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wan tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_ab.txt
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_ab.txt
done
sort $O/train_ab.txt
