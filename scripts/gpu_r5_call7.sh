#!/bin/bash
# Round 5, call 7: the six-wave mid-size GEMM tiles (key 30): operator microbenchmark + bit identity, the in-model A/B at the shapes they are
# dispatched for (512^2 B = 2 CFG -- the reference's example --, 512^2 B = 4, 1024^2 B = 1), the bench shape as a control, the bit-identity tests.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $O
timeout 300 python scripts/mid_tiles_ab.py > $O/mid_tiles_op_ab.txt 2>&1; cat $O/mid_tiles_op_ab.txt | tail -n 8
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --no-kernel-timing"
run() { for v in 0 1 0 1; do MI355_TUNE="30=$v" timeout 300 python bench.py $2 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 key30=$v', d['value'], d['ms_per_step'])" >> $O/mid_tiles_inmodel_ab.txt; done; }
run b2_512_cfg "--steps 20 --warmup 3 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10"
run b4_512 "--steps 20 --warmup 3 --size 512 --batch 4 --denoise-steps 10"
run b1_1024 "--steps 5 --warmup 2 --size 1024 --batch 1"
for v in 0 1; do MI355_TUNE="30=$v" timeout 300 python bench.py --steps 3 --warmup 1 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b8_1024 (control) key30=$v', d['value'], d['ms_per_step'])" >> $O/mid_tiles_inmodel_ab.txt; done
cat $O/mid_tiles_inmodel_ab.txt
( time MI355_TUNE="30=1" timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "mid_size or bitwise or linear or full_size" ) > $O/pytest_mid_tiles.txt 2>&1; echo "rc=$?" >> $O/pytest_mid_tiles.txt
grep -h "passed\|failed\|rc=\|Error" $O/pytest_mid_tiles.txt | tail -n 5
