#!/bin/bash
# Round 6, call 18: what bounds the pipelined dK/dV loop?  Kernel durations (rocprofv3 --kernel-trace --stats) at B = 2, H = 24, S = 4429 for the
# round-3 kernels (43=0), the shipped loop (43=1, transposed reads now from gap 8) and its ablation builds (43=2..5); then the operator test.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_ALLOW_ABLATION=1
O=$GRAFT_REPO_ROOT/gpurun_out/r06r; mkdir -p $O
for v in 0 1 2 3 4 5; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$v -o a -- python $GRAFT_REPO_ROOT/scripts/attn_bwd_ablate.py $v > $O/p$v.log 2>&1)
  python - $v <<'PY' >> $O/ablation.txt
import csv, glob, os, sys
v = sys.argv[1]
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], f"gpurun_out/r06r/p{v}/**/*kernel_stats.csv"), recursive=True)
for r in csv.DictReader(open(f[0])):
    if "attn_bwd_d" in r["Name"] or "attn_kernel" in r["Name"]:
        print(f'43={v} {r["Name"][:90]:90s} calls {int(r["Calls"]):3d} avg_us {float(r["AverageNs"]) / 1e3:8.1f}')
PY
done
cat $O/ablation.txt
unset MI355_ALLOW_ABLATION
( time timeout 300 python -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "attention_backward_matches" ) > $O/pytest_attn_bwd.txt 2>&1; echo "rc=$?" >> $O/pytest_attn_bwd.txt
grep -h "passed\|failed\|rc=\|FAILED\|Error\|assert" $O/pytest_attn_bwd.txt | cut -c1-300 | tail -n 6
