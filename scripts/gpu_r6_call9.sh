#!/bin/bash
# Round 6, call 9: the whole -m gpu suite at this HEAD (timing after the GPU-oracle conversion), smoke(), schedule traces re-recorded, the
# optimize() step with two-round mid-kernel grids allowed (key 37 = 512), and the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06i; mkdir -p $O $O/traces
( time MI355_DUMP_TRACES=$O/traces timeout 1500 python -m pytest tests -q -m gpu --durations=20 ) > $O/pytest_gpu_full.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_full.txt
grep -h "passed\|failed\|rc=\|real\|FAILED" $O/pytest_gpu_full.txt | cut -c1-300 | tail -n 12
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.txt 2>&1; tail -n 4 $O/smoke.txt | cut -c1-300
for t in "37=256" "37=512" "37=256" "37=512"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 train b2_1024 tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_mid_two_rounds.txt
done
cat $O/train_mid_two_rounds.txt
( time timeout 1500 python bench.py 2>$O/bench_default.err ) > $O/bench_b8_ncfg1.json 2> $O/bench_default.time; cut -c1-300 $O/bench_b8_ncfg1.json; tail -n 3 $O/bench_default.time
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06i/bench_b8_ncfg1.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["forward"]["frac"])
for k in ("small_batch","optimize_step","optimize_step_flux1","optimize_step_qwen_image","optimize_step_wan21","vae_decode","families"):
    print(k, json.dumps(d.get(k))[:600])
PY
