#!/bin/bash
# Round 5, call 5: (a) how much of the rollout's speed is data-dependent power (scripts/power_ceiling.py: the same rollout on all-zero weights);
# (b) the driver's own command shape, timed; (c) smoke(); (d) the whole -m gpu suite at the final HEAD.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_ROUND=5
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O
timeout 300 python scripts/power_ceiling.py --scales 1.0,0.0,0.01 > $O/power_ceiling_b8_1024.jsonl 2> $O/power_ceiling.err; cat $O/power_ceiling_b8_1024.jsonl; tail -n 3 $O/power_ceiling.err
timeout 200 python scripts/power_ceiling.py --batch 2 --size 512 --denoise-steps 10 --rollouts 20 --scales 1.0,0.0 > $O/power_ceiling_b2_512.jsonl 2>/dev/null; cat $O/power_ceiling_b2_512.jsonl
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null ) > $O/bench_driver_form.json 2> $O/bench_driver_form.time; cut -c1-300 $O/bench_driver_form.json; tail -n 3 $O/bench_driver_form.time
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
( time timeout 1800 python -m pytest tests -q -m gpu --durations=12 ) > $O/pytest_gpu_full.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_full.txt
grep -h "passed\|failed\|rc=\|^real\|FAILED" $O/pytest_gpu_full.txt | tail -n 8
