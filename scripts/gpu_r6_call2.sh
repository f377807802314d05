#!/bin/bash
# Round 6, call 2: first contact of the mid-size GEMM kernel (128x192 / 192x128 tiles, key 32): operator-level bit-identity + timing,
# whole-forward bit-identity, then the in-model A/B at the reference's 512^2 B = 2 CFG example shape and at 1024^2 B = 1 / 2.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06b; mkdir -p $O
timeout 600 python scripts/gemm_mid_ab.py > $O/gemm_mid_ab.txt 2>&1; echo "rc=$?" >> $O/gemm_mid_ab.txt
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-families --no-train-step --no-vae"
for m in 0 1 2 0 1; do
  MI355_TUNE="32=$m" timeout 300 python bench.py --steps 20 --warmup 3 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b2_512_cfg mid=$m', d['value'], d['ms_per_step'])" >> $O/inmodel_ab.txt
done
for m in 0 1 2; do
  MI355_TUNE="32=$m" timeout 300 python bench.py --steps 5 --warmup 2 --size 1024 --batch 1 --denoise-steps 28 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b1_1024 mid=$m', d['value'], d['ms_per_step'])" >> $O/inmodel_ab.txt
  MI355_TUNE="32=$m" timeout 300 python bench.py --steps 5 --warmup 2 --size 1024 --batch 2 --denoise-steps 28 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b2_1024 mid=$m', d['value'], d['ms_per_step'])" >> $O/inmodel_ab.txt
  MI355_TUNE="32=$m" timeout 300 python bench.py --steps 3 --warmup 1 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b8_1024 mid=$m', d['value'], d['ms_per_step'])" >> $O/inmodel_ab.txt
done
cat $O/gemm_mid_ab.txt | cut -c1-700 | tail -n 40
cat $O/inmodel_ab.txt
