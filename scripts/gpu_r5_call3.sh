#!/bin/bash
# Round 5, call 3: the tests written / changed since call 2 first (parity fixes, native evaluation-mode Wan sampling, UniPC kernels), then the whole
# -m gpu suite at this HEAD (the library changed under every family: LayerNorm widths, gradient-buffer registry, Wan scratch).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_wan.py tests/test_gpu_wan_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_fullsize.py -q -s -m gpu -k "evaluation_mode or unipc or one_block or 40_head or advantages or replay_gradients_vs_oracle" ) > $O/pytest_new.txt 2>&1; echo "rc=$?" >> $O/pytest_new.txt
grep -h "passed\|failed\|rc=\|Error\|worst\|advantages (M\|log-prob\|evaluation-mode" $O/pytest_new.txt | cut -c1-1500 | tail -n 30
( time timeout 1800 python -m pytest tests -q -m gpu --durations=15 ) > $O/pytest_gpu_full.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_full.txt
grep -h "passed\|failed\|rc=\|^real\|FAILED" $O/pytest_gpu_full.txt | tail -n 12
