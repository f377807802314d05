#!/bin/bash
# Round 6, call 26: head_dim-128 attention backward with the tail / key masks peeled into the last tile's copy of the body: backward suites of the
# three head_dim-128 families, then their optimize() steps (2 runs each).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06x; mkdir -p $O
( time timeout 1800 python -m pytest tests/test_gpu_flux_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_wan_backward.py -x -q -m gpu ) > $O/pytest_backward_128.txt 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest_backward_128.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error" $O/pytest_backward_128.txt | cut -c1-300 | tail -n 8
if [ $rc -ne 0 ]; then exit 0; fi
for i in 1 2; do
  timeout 300 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('flux1', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train.txt
  timeout 300 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('qwen', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train.txt
  timeout 300 python scripts/wan_train_bench.py --batch 1 --frames 49 --iters 2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wan', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train.txt
done
sort $O/train.txt
