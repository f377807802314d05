#!/bin/bash
# round 4, third GPU call: FLUX.1 backward at full width + its step timing / profile, the RCCL test, the new bench.py legs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flux_backward.py -q -s -m gpu -k "full_width" > $O/pytest_flux_full_width.txt 2>&1; echo "rc=$?" >> $O/pytest_flux_full_width.txt
timeout 300 python -m pytest tests/test_gpu_ddp_rccl.py -x -q -s -m gpu > $O/pytest_ddp_rccl.txt 2>&1; echo "rc=$?" >> $O/pytest_ddp_rccl.txt
timeout 600 python scripts/flux_train_bench.py --batch 1 --size 1024 > $O/flux_train_b1_1024.json 2> $O/flux_train_b1_1024.err
timeout 600 python scripts/flux_train_bench.py --batch 1 --size 384 > $O/flux_train_b1_384.json 2> $O/flux_train_b1_384.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_flux_train -o t -- python $GRAFT_REPO_ROOT/scripts/flux_train_bench.py --only-step --iters 1 > $O/prof_flux_train.log 2>&1)
python - <<'P' > $O/flux_train_kernel_stats.txt 2>&1
import csv, glob, os
f = sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r04c/prof_flux_train/**/*kernel_stats*.csv"), recursive=True))
rows = list(csv.DictReader(open(f[0])))
print(f"{'kernel':110s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[:40]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
P
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
find $O -type f -size +1M -delete
tail -n 4 $O/*.txt $O/*.json | cut -c1-1500
