#!/bin/bash
# Round 5, call 2: (a) measured CEILINGS of the forward fusions VERDICT r4 asks for, before building any of them: the rollout with the launches a
# fusion would absorb simply SKIPPED (mi355_tune_set key 29; wrong results, right timing) -- text chain (grouped image + text launch), every
# LayerNorm-modulate (GEMM-prologue fusion), V^T projection (fused q|k|v weight) -- at the bench shape and at the reference's 512^2 B = 2 CFG example
# shape; (b) the tightened parity tests (advantages with well-conditioned rewards, one-block gradient VALUES for Wan / Qwen-Image, the 40-head Wan width).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; mkdir -p $O
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --no-kernel-timing"
for m in 0 1 2 4 7; do
  MI355_TUNE="29=$m" timeout 300 python bench.py --steps 3 --warmup 1 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b8_1024 ablate=$m', d['value'], d['ms_per_step'])" >> $O/ablate.txt
done
for m in 0 1 2 4 7; do
  MI355_TUNE="29=$m" timeout 300 python bench.py --steps 20 --warmup 3 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b2_512_cfg ablate=$m', d['value'], d['ms_per_step'])" >> $O/ablate.txt
done
cat $O/ablate.txt
( time timeout 900 python -m pytest tests/test_gpu_wan_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_fullsize.py -q -s -m gpu -k "one_block or 40_head or advantages" ) > $O/pytest_parity.txt 2>&1; echo "rc=$?" >> $O/pytest_parity.txt
grep -h "passed\|failed\|rc=\|Error\|worst\|advantages\|log-prob\|assert" $O/pytest_parity.txt | cut -c1-1500 | tail -n 30
