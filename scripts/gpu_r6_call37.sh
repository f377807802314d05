#!/bin/bash
# Round 6, call 37: head_dim-128 pipelined backward with the masks only in the last tile's bodies: operator test, the three families' backward
# suites, optimize() steps (2 runs each).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06ae; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_flux_backward.py -x -q -m gpu -k "attention128_backward_matches" ) > $O/pytest_attn128_bwd.txt 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest_attn128_bwd.txt
grep -h "passed\|failed\|rc=\|FAILED\|Error\|assert" $O/pytest_attn128_bwd.txt | cut -c1-300 | tail -n 8
if [ $rc -ne 0 ]; then exit 0; fi
( time timeout 1800 python -m pytest tests/test_gpu_flux_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_wan_backward.py -x -q -m gpu ) > $O/pytest_backward_128.txt 2>&1; echo "rc=$?" >> $O/pytest_backward_128.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error" $O/pytest_backward_128.txt | cut -c1-300 | tail -n 6
for i in 1 2; do
  timeout 300 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('flux1', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train.txt
  timeout 300 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('qwen', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train.txt
  timeout 300 python scripts/wan_train_bench.py --batch 1 --frames 49 --iters 2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wan', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train.txt
  MI355_TUNE="44=0" timeout 300 python scripts/wan_train_bench.py --batch 1 --frames 49 --iters 2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wan_round4_kernels', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train.txt
done
sort $O/train.txt
