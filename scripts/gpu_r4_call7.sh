#!/bin/bash
# round 4, seventh GPU call: bf16 gradient buffers written by the reduction kernels (bit-identity vs the fp32 route, three families), the Qwen-Image
# gradient tests on the conditioned full-width model, SD3 backward regression, and the optimize()-step timings of the three families with bf16 masters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16_grad_buffers.py -q -s -m gpu > $O/pytest_bf16_grads.txt 2>&1; echo "rc=$?" >> $O/pytest_bf16_grads.txt
timeout 900 python -m pytest tests/test_gpu_qwen_backward.py tests/test_gpu_qwen.py -q -s -m gpu > $O/pytest_qwen.txt 2>&1; echo "rc=$?" >> $O/pytest_qwen.txt
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -m gpu > $O/pytest_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_backward.txt
timeout 600 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 3 > $O/train_bench_attn.json 2>/dev/null
timeout 400 python scripts/flux_train_bench.py --batch 1 --size 1024 --iters 2 > $O/flux_train_bench.json 2>/dev/null
timeout 700 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 2 > $O/qwen_train_bench.json 2>/dev/null
(cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_qwen -o t -- python $GRAFT_REPO_ROOT/scripts/qwen_train_bench.py --only-step --iters 2 > $O/prof_qwen.log 2>&1)
python - <<'P' > $O/qwen_train_step_kernel_stats.txt 2>&1
import csv, glob, os
f = sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04g/prof_qwen/**/*kernel_stats*.csv"), recursive=True))
rows = list(csv.DictReader(open(f[0])))
for r in rows[:32]:
    print(f"{r['Name'][:100]:100s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):5.1f}%")
P
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rel-L2\|rc=\|Error\|bit-identical" $O/pytest_*.txt | cut -c1-400 | tail -n 30
tail -n 2 $O/*.json | cut -c1-1200
head -n 24 $O/qwen_train_step_kernel_stats.txt | cut -c1-170
