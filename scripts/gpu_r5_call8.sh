#!/bin/bash
# Round 5, call 8: SD3.5 optimize() step on the reference's default target set: weight-gradient GEMMs on a side stream (key 26 = 2, opt-in since
# round 4 where it measured neutral on the 349 M-parameter set) A/B -- keep as default or delete.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O
for v in 1 2 1 2; do MI355_TUNE="26=$v" timeout 200 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 5 --only-step 2>/dev/null | sed "s/^/key26=$v /" >> $O/sd3_wgrad_side_ab.txt; done
for v in 1 2; do MI355_TUNE="26=$v" timeout 200 python scripts/train_bench.py --batch 2 --size 512 --train default --iters 8 --only-step 2>/dev/null | sed "s/^/512 key26=$v /" >> $O/sd3_wgrad_side_ab.txt; done
cat $O/sd3_wgrad_side_ab.txt
