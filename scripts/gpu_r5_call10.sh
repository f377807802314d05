#!/bin/bash
# Round 5, call 10: dispatch / schedule knob sweep at the reference's 512^2 B = 2 CFG example shape (not power-bound: DESIGN 14.8 -- schedule choices
# pay one for one there) and at B = 2 1024^2; each setting twice, baseline interleaved.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $O
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --no-kernel-timing"
one() { MI355_TUNE="$2" timeout 200 python bench.py $3 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 tune=[$2]', d['value'])" >> $O/knob_sweep.txt; }
S512="--steps 30 --warmup 4 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10"
for t in "" "3=96" "3=64" "3=512" "8=0" "10=1" "10=0" "11=1" "" "0=3" "0=2" "19=8192" "7=0" "7=3" "7=12" "1=0" "6=0" ""; do one b2_512_cfg "$t" "$S512"; done
S1024="--steps 6 --warmup 2 --size 1024 --batch 2"
for t in "" "3=96" "3=192" "10=1" "11=1" "0=3" "0=2" "19=8192" ""; do one b2_1024 "$t" "$S1024"; done
cat $O/knob_sweep.txt
