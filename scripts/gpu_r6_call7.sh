#!/bin/bash
# Round 6, call 7: the weight-gradient GEMM on row-major operands (gemm_tn.hip, key 39) -- operator bit-identity vs the transposed-copy path,
# in-engine gradients, the gradient tests of all families (transpose tile swizzle), then the optimize() step A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06g; mkdir -p $O
( time timeout 1200 python -m pytest -q -s -m gpu --durations=8 tests/test_gpu_backward.py tests/test_gpu_bf16_grad_buffers.py tests/test_gpu_fullsize.py tests/test_gpu_flux_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_wan_backward.py tests/test_gpu_grpo_epoch.py tests/test_gpu_ddp_rccl.py ) > $O/pytest_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_backward.txt
grep -h "passed\|failed\|rc=\|Error\|real\|FAILED" $O/pytest_backward.txt | cut -c1-300 | tail -n 20
for t in "39=1" "39=0" "39=1" "39=0"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train b2_1024 tune=$t', d['ms_forward_backward'], d['ms_forward_train'], d['frac_of_2500'])" >> $O/train_ab.txt
done
for t in "39=1" "39=0"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 512 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train b2_512 tune=$t', d['ms_forward_backward'])" >> $O/train_ab.txt
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train blocks --iters 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train b2_1024 all-block-linears tune=$t', d['ms_forward_backward'])" >> $O/train_ab.txt
done
cat $O/train_ab.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train default --iters 3 --only-step > $O/prof_train.log 2>&1)
python - <<'PY' > $O/sd3_train_step_kernel_stats.txt 2>&1
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06g/prof_train/**/*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:34]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f} {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
PY
grep '^{' $O/prof_train.log >> $O/sd3_train_step_kernel_stats.txt; find $O -type f -size +2M -delete; head -n 26 $O/sd3_train_step_kernel_stats.txt | cut -c1-180
