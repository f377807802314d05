#!/bin/bash
# Round 6, call 17 (after zeroing the padding rows of dO | -L | -Delta in the prologue kernel): second run of the software-pipelined attention-backward passes (gen_attn_bwd64.py): the operator test (bit identity with
# the round-3 kernels at 1..70 tiles), then the backward suite, the optimize() step A/B over key 43 and a rocprof stats pass.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06q; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "attention_backward_matches" ) > $O/pytest_attn_bwd.txt 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest_attn_bwd.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error\|assert" $O/pytest_attn_bwd.txt | cut -c1-300 | tail -n 14
if [ $rc -ne 0 ]; then exit 0; fi
( time timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16_grad_buffers.py -x -q -m gpu ) > $O/pytest_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_backward.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error" $O/pytest_backward.txt | cut -c1-300 | tail -n 8
for i in 1 2 3; do for t in "43=1" "43=0"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 train b2_1024 tune=$t', d['ms_forward_backward'], d['ms_forward_train'], d.get('ms_forward_nograd'))" >> $O/train_ab.txt
done; done
cat $O/train_ab.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train default --iters 3 --only-step > $O/prof_train.log 2>&1)
python - <<'PY' > $O/train_kernel_stats.txt 2>&1
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06q/prof_train/**/*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f} {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
PY
grep '^{' $O/prof_train.log >> $O/train_kernel_stats.txt; head -n 6 $O/train_kernel_stats.txt | cut -c1-200
