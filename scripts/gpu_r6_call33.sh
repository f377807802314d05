#!/bin/bash
# Round 6, call 33: first contact of the two-row-block head_dim-64 attention-backward passes (gen_attn_bwd64x2.py, key 43 = 6): operator test (bit
# identity at 1..70 tiles), then the SD3.5 backward tests and the optimize() step under 43 = 1 / 6, kernel durations through rocprof.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06ac; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "attention_backward_matches" ) > $O/pytest_attn_bwd.txt 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest_attn_bwd.txt
grep -h "passed\|failed\|rc=\|FAILED\|Error\|assert" $O/pytest_attn_bwd.txt | cut -c1-300 | tail -n 10
if [ $rc -ne 0 ]; then exit 0; fi
( time MI355_TUNE="43=6" timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16_grad_buffers.py -x -q -m gpu ) > $O/pytest_backward_x2.txt 2>&1; echo "rc=$?" >> $O/pytest_backward_x2.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error" $O/pytest_backward_x2.txt | cut -c1-300 | tail -n 6
for i in 1 2 3; do for t in "43=1" "43=6"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 train b2_1024 tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_ab.txt
done; done
sort $O/train_ab.txt
for v in 1 6; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$v -o a -- python $GRAFT_REPO_ROOT/scripts/attn_bwd_ablate.py $v > $O/p$v.log 2>&1)
  python - $v <<'PY' >> $O/kernel_us.txt
import csv, glob, os, sys
v = sys.argv[1]
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], f"gpurun_out/r06ac/p{v}/**/*kernel_stats.csv"), recursive=True)
for r in csv.DictReader(open(f[0])):
    if "attn_bwd_d" in r["Name"]:
        print(f'43={v} {r["Name"][:90]:90s} calls {int(r["Calls"]):3d} avg_us {float(r["AverageNs"]) / 1e3:8.1f}')
PY
done
cat $O/kernel_us.txt
