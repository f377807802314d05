#!/bin/bash
# FIRST GPU call of round 6 (VERDICT r5 "next round" #1, tests only, no new kernel): the headline configurations compared with the fp32 oracle
# AT THEIR OWN LENGTH / DEPTH, the oracle running on the GPU in fp32 (tests/_gpu_oracle.py):
#   (a) tests/test_gpu_fullsize.py  -- config B at N = 28 (per-step latents, log-prob, oracle replay), config A, the CFG pair, advantages, gradients
#   (b) tests/test_gpu_full_depth.py -- FLUX.1-dev 19+38 blocks, Wan2.1-1.3B 30 blocks at 20 280 tokens, Qwen-Image 60 layers at 1328^2
#   (c) the whole -m gpu suite with durations (the suite was at 814 s of the driver's 1200 s: which tests to trim)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06a; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -m gpu --durations=0 ) > $O/pytest_fullsize.txt 2>&1; echo "rc=$?" >> $O/pytest_fullsize.txt
( time timeout 900 python -m pytest tests/test_gpu_full_depth.py -q -s -m gpu --durations=0 ) > $O/pytest_full_depth.txt 2>&1; echo "rc=$?" >> $O/pytest_full_depth.txt
( time timeout 1500 python -m pytest tests -q -m gpu --durations=60 --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_full_depth.py ) > $O/pytest_rest.txt 2>&1; echo "rc=$?" >> $O/pytest_rest.txt
grep -h "config \|full depth\|worst\|passed\|failed\|rc=\|Error\|error\|real" $O/pytest_fullsize.txt $O/pytest_full_depth.txt | cut -c1-420 | tail -n 90
tail -n 75 $O/pytest_rest.txt | cut -c1-200
