#!/bin/bash
# round 4, last GPU call: the driver's own bench command at HEAD (wall clock of the whole line incl. the families / optimize() legs), and the
# Qwen-Image optimize() step at BASELINE.json configs[4]'s own shape (1328^2, B = 2, true CFG: 4 x 6985 tokens) -- does it fit one MI355X?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04o; mkdir -p $O
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_driver_form.err ) > $O/bench_driver_form.json 2> $O/bench_driver_form.time
timeout 600 python scripts/qwen_train_bench.py --batch 2 --size 1328 --n-text 64 --iters 1 > $O/qwen_train_b2_1328.json 2> $O/qwen_train_b2_1328.err; echo "rc=$?" >> $O/qwen_train_b2_1328.err
timeout 600 python scripts/qwen_train_bench.py --batch 1 --size 1328 --n-text 64 --iters 2 > $O/qwen_train_b1_1328.json 2> $O/qwen_train_b1_1328.err; echo "rc=$?" >> $O/qwen_train_b1_1328.err
find $O -type f -size +1M -delete
tail -n 3 $O/bench_driver_form.time; cut -c1-300 $O/bench_driver_form.json
tail -n 2 $O/qwen_train_b2_1328.json $O/qwen_train_b2_1328.err $O/qwen_train_b1_1328.json $O/qwen_train_b1_1328.err | cut -c1-900
