#!/bin/bash
# Round 5, call 12: dispatch-key sweep on the SD3.5 optimize() step (B = 2, 1024^2 and 512^2, default target set): bit-identical kernel choices only.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05l; mkdir -p $O
for t in "" "0=3" "0=2" "3=96" "3=192" "3=256" "31=256" "31=1024" "19=8192" "7=3" "7=12" ""; do
  MI355_TUNE="$t" timeout 100 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 5 --only-step 2>/dev/null | sed "s/^/b2_1024 tune=[$t] /" >> $O/train_knob_sweep.txt
done
for t in "" "0=3" "3=96" "3=256" "31=256" "7=3" ""; do
  MI355_TUNE="$t" timeout 100 python scripts/train_bench.py --batch 2 --size 512 --train default --iters 8 --only-step 2>/dev/null | sed "s/^/b2_512 tune=[$t] /" >> $O/train_knob_sweep.txt
done
cat $O/train_knob_sweep.txt
