#!/bin/bash
# round 4: per-kernel tables of the three optimize() steps at HEAD (rocprofv3 --kernel-trace --stats of 1 + 2 steps each)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04p; mkdir -p $O
prof() {  # tag, script, args...
  tag=$1; shift
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/"$@" > $O/prof_$tag.log 2>&1)
  python - "$O/prof_$tag" > $O/${tag}_train_step_kernel_stats.txt 2>&1 <<'P'
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats*.csv"), recursive=True))
rows = list(csv.DictReader(open(f[0])))
for r in rows[:36]:
    print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):5.1f}%")
P
  grep '^{' $O/prof_$tag.log | tail -n 1 >> $O/${tag}_train_step_kernel_stats.txt
}
prof sd3 train_bench.py --batch 2 --size 1024 --train attn --only-step --iters 2
prof flux flux_train_bench.py --batch 1 --size 1024 --only-step --iters 2
prof qwen qwen_train_bench.py --batch 1 --size 1024 --only-step --iters 2
find $O -type f -size +1M -delete
head -n 14 $O/sd3_train_step_kernel_stats.txt | cut -c1-175; head -n 12 $O/flux_train_step_kernel_stats.txt | cut -c1-175
