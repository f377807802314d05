#!/bin/bash
# Round 6, call 8: after the DMOD = false ln_mod_bwd instantiations and the 4-deep column-sum loads: backward tests of every family, optimize() steps.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06h; mkdir -p $O
( time timeout 1200 python -m pytest -q -m gpu --durations=5 tests/test_gpu_backward.py tests/test_gpu_bf16_grad_buffers.py tests/test_gpu_fullsize.py tests/test_gpu_flux_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_wan_backward.py tests/test_gpu_grpo_epoch.py tests/test_gpu_ddp_rccl.py tests/test_gpu_schedules.py ) > $O/pytest_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_backward.txt
grep -h "passed\|failed\|rc=\|Error\|real\|FAILED" $O/pytest_backward.txt | cut -c1-300 | tail -n 12
for i in 1 2; do
  timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 train b2_1024', d['ms_forward_backward'], d['ms_forward_train'], d['frac_of_2500'])" >> $O/train.txt
done
timeout 300 python scripts/train_bench.py --batch 2 --size 512 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 train b2_512', d['ms_forward_backward'])" >> $O/train.txt
timeout 600 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 2 2>/dev/null | tail -n 1 | cut -c1-400 >> $O/train.txt
timeout 600 python scripts/wan_train_bench.py --batch 1 --iters 2 2>/dev/null | tail -n 1 | cut -c1-400 >> $O/train.txt
cat $O/train.txt
