#!/bin/bash
# Round 6, call 13: a longer interleaved A/B of the optimize() step over key 39 (1 = 256x256-tile wgrad where it fits, 2 = 128x128 only): call 12's
# five runs differed by less than their run-to-run spread.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06m; mkdir -p $O
for i in 1 2 3 4 5 6; do for t in "39=1" "39=2"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 12 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 train b2_1024 tune=$t', d['ms_forward_backward'], d['ms_forward_train'])" >> $O/train_tn256_ab.txt
done; done
cat $O/train_tn256_ab.txt
python - <<'PY'
import statistics as st
r = {}
for l in open("gpurun_out/r06m/train_tn256_ab.txt"):
    p = l.split(); r.setdefault(p[3], []).append(float(p[4]))
for k, v in r.items(): print(k, "median", st.median(v), "min", min(v), "mean", round(st.mean(v), 2))
PY
