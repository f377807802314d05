#!/bin/bash
# Round 6, call 22: per-kernel tables of the Wan2.1 (config D: 20 280 tokens, CFG) and FLUX.1 optimize() steps (where does the time go at head_dim 128?).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06v; mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wan -o train -- python $GRAFT_REPO_ROOT/scripts/wan_train_bench.py --batch 1 --frames 49 --iters 2 > $O/prof_wan.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_flux -o train -- python $GRAFT_REPO_ROOT/scripts/flux_train_bench.py --batch 1 --size 1024 --iters 2 > $O/prof_flux.log 2>&1)
for fam in wan flux; do
python - $fam <<'PY' > $O/${fam}_train_step_kernel_stats.txt 2>&1
import csv, glob, os, sys
fam = sys.argv[1]
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), f"gpurun_out/r06v/prof_{fam}/**/*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:120]:120s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f} {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
PY
grep '^{' $O/prof_$fam.log | cut -c1-400 >> $O/${fam}_train_step_kernel_stats.txt
head -n 16 $O/${fam}_train_step_kernel_stats.txt | cut -c1-220
done
