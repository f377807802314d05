#!/bin/bash
# Round 6, call 30: first contact of the software-pipelined head_dim-128 attention-backward passes (gen_attn_bwd128.py): the operator test (bit
# identity with the round-4 kernels at 1..72 tiles), the backward suites of the three head_dim-128 families, their optimize() steps under
# key 44 = 1 / 0, and a rocprof stats pass of the Wan step.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06aa; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_flux_backward.py -x -q -m gpu -k "attention128_backward_matches" ) > $O/pytest_attn128_bwd.txt 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest_attn128_bwd.txt
grep -h "passed\|failed\|rc=\|FAILED\|Error\|assert" $O/pytest_attn128_bwd.txt | cut -c1-300 | tail -n 10
if [ $rc -ne 0 ]; then exit 0; fi
( time timeout 1800 python -m pytest tests/test_gpu_flux_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_wan_backward.py -x -q -m gpu ) > $O/pytest_backward_128.txt 2>&1; echo "rc=$?" >> $O/pytest_backward_128.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error" $O/pytest_backward_128.txt | cut -c1-300 | tail -n 8
for t in "44=1" "44=0" "44=1" "44=0"; do
  MI355_TUNE="$t" timeout 300 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('flux1 tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_ab.txt
  MI355_TUNE="$t" timeout 300 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('qwen tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_ab.txt
  MI355_TUNE="$t" timeout 300 python scripts/wan_train_bench.py --batch 1 --frames 49 --iters 2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wan tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_ab.txt
done
sort $O/train_ab.txt
