#!/bin/bash
# Round 6, call 12: the 256x256-tile row-major weight-gradient GEMM (gemm_tn256_kernel) and in-kernel bias column sums: backward tests, an
# op-level timing of the three forms at the SD3.5 shape, the optimize() step A/B over key 39 (1 = 256-tile where it fits, 2 = 128-tile only,
# 0 = transposed copies), and a rocprof stats pass of the step.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06l; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16_grad_buffers.py -x -q -m gpu ) > $O/pytest_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_backward.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error" $O/pytest_backward.txt | cut -c1-300 | tail -n 12
timeout 300 python - > $O/op_wgrad_timing.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "flow-factory_amd")
from mi355_flow import engine
torch.manual_seed(0)
for (M, N, K, splits) in ((8192, 1536, 1536, (4, 7)), (8192, 1536, 6144, (1, 2)), (16384, 1536, 1536, (7,)), (4096, 1536, 1536, (7,))):
    dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, K, device="cuda").bfloat16()
    for split in splits:
        for variant in (0, 1, 2):
            for _ in range(3): engine.op_wgrad(dy, x, split, variant=variant)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): engine.op_wgrad(dy, x, split, variant=variant)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"M={M} N={N} K={K} split={split} variant={variant}: {us:.1f} us/call (incl. reduce), {2*M*N*K/us/1e6:.0f} TFLOP/s")
PY
cat $O/op_wgrad_timing.txt | tail -n 30
for t in "39=1" "39=2" "39=0" "39=1" "39=2"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 train b2_1024 tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_tn256_ab.txt
done
cat $O/train_tn256_ab.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train default --iters 3 --only-step > $O/prof_train.log 2>&1)
python - <<'PY' > $O/train_kernel_stats.txt 2>&1
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06l/prof_train/**/*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f} {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
PY
grep '^{' $O/prof_train.log >> $O/train_kernel_stats.txt; head -n 24 $O/train_kernel_stats.txt | cut -c1-200
