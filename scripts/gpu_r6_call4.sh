#!/bin/bash
# Round 6, call 4: (a) the tests converted to the GPU oracle / band-derived tolerances (first run), (b) which launch classes of the mid-size kernel
# pay in-model (mask A/B, two-stream and single-stream), (c) bench.py's optimize_step_ddp leg on a world-size-1 RCCL group.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $O
( time timeout 1200 python -m pytest -q -s -m gpu --durations=25 tests/test_gpu_model.py tests/test_gpu_flux.py tests/test_gpu_wan.py tests/test_gpu_qwen.py tests/test_gpu_wan_backward.py tests/test_gpu_qwen_backward.py tests/test_gpu_flux_backward.py tests/test_gpu_schedules.py ) > $O/pytest_converted.txt 2>&1; echo "rc=$?" >> $O/pytest_converted.txt
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-families --no-train-step --no-vae"
for t in "32=0" "32=1,37=256,36=1" "32=1,37=256,36=2" "32=1,37=256,36=4" "32=1,37=256,36=8" "32=1,37=256" "32=0,8=0" "32=1,37=256,8=0"; do
  MI355_TUNE="$t" timeout 300 python bench.py --steps 20 --warmup 3 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b2_512_cfg tune=$t', d['value'], d['ms_per_step'])" >> $O/inmodel_mask_ab.txt
done
for t in "32=0" "32=1,37=256,36=1" "32=1,37=256,36=2" "32=1,37=256,36=4" "32=1,37=256" "32=0,8=0" "32=1,37=256,8=0"; do
  MI355_TUNE="$t" timeout 300 python bench.py --steps 5 --warmup 2 --size 1024 --batch 1 --denoise-steps 28 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b1_1024 tune=$t', d['value'], d['ms_per_step'])" >> $O/inmodel_mask_ab.txt
done
timeout 600 python bench.py --steps 1 --warmup 1 --ddp-step-world1 $COMMON > $O/bench_ddp_world1.json 2> $O/bench_ddp_world1.err; echo "rc=$?" >> $O/bench_ddp_world1.err
grep -h "passed\|failed\|rc=\|Error\|real\|band" $O/pytest_converted.txt | cut -c1-330 | tail -n 60
cat $O/inmodel_mask_ab.txt
python -c "import json; d=json.loads([l for l in open('$O/bench_ddp_world1.json') if l.startswith('{')][-1]); print(json.dumps(d.get('optimize_step_ddp'), indent=1))"; tail -n 3 $O/bench_ddp_world1.err
