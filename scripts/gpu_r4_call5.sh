#!/bin/bash
# round 4, fifth GPU call: d = 64 attention backward after the batched-read / branch-free change (numerics + SD3 optimize step), the 48-channel Wan
# forward (TI2V geometry), recorded schedule traces for the CPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -m gpu > $O/pytest_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_backward.txt
timeout 600 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 3 > $O/train_bench_attn.json 2>/dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --only-step --iters 2 > $O/prof_train.log 2>&1)
python - <<'P' > $O/train_step_kernel_stats.txt 2>&1
import csv, glob, os
f = sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04e/prof_train/**/*kernel_stats*.csv"), recursive=True))
rows = list(csv.DictReader(open(f[0])))
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):5.1f}%")
P
timeout 300 python -m pytest tests/test_gpu_wan.py -x -q -m gpu -k "48_latent or forward_matches" > $O/pytest_wan48.txt 2>&1; echo "rc=$?" >> $O/pytest_wan48.txt
MI355_DUMP_TRACES=$O/traces timeout 600 python -m pytest tests/test_gpu_schedules.py -x -q -s -m gpu -k "race_free" > $O/pytest_schedules.txt 2>&1; echo "rc=$?" >> $O/pytest_schedules.txt
find $O -type f -size +1M -delete
tail -n 5 $O/*.txt $O/*.json | cut -c1-1200
