#!/bin/bash
# Round 5, call 14: experiment -- the K = 6144 gated-residual GEMM (second MLP linear) of 4096-row grids on the ping-pong kernel (96 tiles) instead of the
# 128 x 128 generic kernel (key 32 = K threshold); A/B at the 512^2 example shapes and 1024^2 B = 1.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; mkdir -p $O
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --no-kernel-timing"
one() { MI355_TUNE="$2" timeout 200 python bench.py $3 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 tune=[$2]', d['value'])" >> $O/pp_long_k_ab.txt; }
for t in "" "32=4096" "32=1536" "" "32=4096" "32=1536"; do one b2_512_cfg "$t" "--steps 30 --warmup 4 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10"; done
for t in "" "32=4096" "32=1536"; do one b4_512 "$t" "--steps 30 --warmup 4 --size 512 --batch 4 --denoise-steps 10"; done
for t in "" "32=4096"; do one b1_1024 "$t" "--steps 5 --warmup 2 --size 1024 --batch 1"; done
cat $O/pp_long_k_ab.txt
