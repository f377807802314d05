#!/bin/bash
# round 4, eighth GPU call: Qwen-Image training forward unified with the two-stream no-grad forward (forward_core on the per-block stash), bf16
# gradient buffers on the Qwen engine, step timing + kernel profile at 60 layers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16_grad_buffers.py -q -s -m gpu > $O/pytest_bf16_grads.txt 2>&1; echo "rc=$?" >> $O/pytest_bf16_grads.txt
timeout 900 python -m pytest tests/test_gpu_qwen_backward.py -q -s -m gpu > $O/pytest_qwen_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_qwen_backward.txt
timeout 700 python scripts/qwen_train_bench.py --batch 1 --size 1024 --iters 2 > $O/qwen_train_bench.json 2> $O/qwen_train_bench.err
(cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_qwen -o t -- python $GRAFT_REPO_ROOT/scripts/qwen_train_bench.py --only-step --iters 2 > $O/prof_qwen.log 2>&1)
python - <<'P' > $O/qwen_train_step_kernel_stats.txt 2>&1
import csv, glob, os
f = sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04h/prof_qwen/**/*kernel_stats*.csv"), recursive=True))
rows = list(csv.DictReader(open(f[0])))
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):5.1f}%")
P
find $O -type f -size +1M -delete
grep -h "passed\|failed\|rel-L2\|rc=\|Error\|bit-identical" $O/pytest_*.txt | cut -c1-300 | tail -n 20
tail -n 2 $O/*.json $O/*.err | cut -c1-1200
head -n 40 $O/qwen_train_step_kernel_stats.txt | cut -c1-170
