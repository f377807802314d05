#!/bin/bash
# Round 6, call 5: (a) new tests (fused split-K reduce + column-sum finish, slab gradient buffers, gradient-noise isolation, lincomb promoted
# dtypes), (b) optimize() step with / without the fused finish, (c) the shipped mid-size-kernel rule in-model, (d) PMC counters at the weak shapes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $O
( time timeout 1200 python -m pytest -q -s -m gpu --durations=15 tests/test_gpu_bf16_grad_buffers.py tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_ddp_rccl.py tests/test_gpu_grpo_epoch.py tests/test_gpu_kernels.py tests/test_gpu_adapter.py "tests/test_gpu_wan.py::test_unipc_kernels_match_their_torch_statements" "tests/test_gpu_wan.py::test_wan_evaluation_mode_sampling_matches_oracle" ) > $O/pytest_new.txt 2>&1; echo "rc=$?" >> $O/pytest_new.txt
for t in "38=1" "38=0" "38=1" "38=0"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train b2_1024 tune=$t', d['ms_forward_backward'], d['ms_forward_train'], d['frac_of_2500'])" >> $O/train_ab.txt
done
MI355_TUNE="38=1" timeout 300 python scripts/train_bench.py --batch 2 --size 512 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train b2_512 fused', d['ms_forward_backward'])" >> $O/train_ab.txt
MI355_TUNE="38=0" timeout 300 python scripts/train_bench.py --batch 2 --size 512 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train b2_512 two-launch', d['ms_forward_backward'])" >> $O/train_ab.txt
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-families --no-train-step --no-vae"
for t in "32=0" "32=1" "32=0" "32=1"; do
  MI355_TUNE="$t" timeout 300 python bench.py --steps 20 --warmup 3 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b2_512_cfg tune=$t', d['value'], d['ms_per_step'])" >> $O/inmodel_rule.txt
done
for t in "32=0" "32=1"; do
  MI355_TUNE="$t" timeout 300 python bench.py --steps 5 --warmup 2 --size 1024 --batch 1 --denoise-steps 28 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b1_1024 tune=$t', d['value'], d['ms_per_step'])" >> $O/inmodel_rule.txt
  MI355_TUNE="$t" timeout 300 python bench.py --steps 5 --warmup 2 --size 512 --batch 8 --denoise-steps 10 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b8_512_nocfg tune=$t', d['value'], d['ms_per_step'])" >> $O/inmodel_rule.txt
  MI355_TUNE="$t" timeout 300 python bench.py --steps 5 --warmup 2 --size 512 --batch 4 --guidance 4.5 --denoise-steps 10 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b4_512_cfg tune=$t', d['value'], d['ms_per_step'])" >> $O/inmodel_rule.txt
done
grep -h "passed\|failed\|rc=\|Error\|real\|gradient noise\|bit-identical" $O/pytest_new.txt | cut -c1-420 | tail -n 30
cat $O/train_ab.txt $O/inmodel_rule.txt
MID=1 bash scripts/gpu_r6_pmc.sh > $O/pmc_run.log 2>&1; tail -n 130 $O/pmc_run.log | cut -c1-250
