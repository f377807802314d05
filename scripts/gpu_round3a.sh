# First GPU call of round 3 (everything here was written after round 2's GPU budget was spent and has NOT run on an MI355X yet):
#   1. the gated checks (tests/test_gpu_next_round.py): SD3.5-large width forward vs the oracle; bench.py's small-batch legs; two-stream /
#      graph-replay bit-identity for FLUX.1, Qwen-Image, Wan; attention variant 3
#   2. the delivered-clock probe beside the hot kernels (scripts/clock_under_load.py) -> which kernel still has headroom at ITS clock
#   3. the whole -m gpu suite at HEAD
#   4. A/B of the Qwen-Image (keys 12, 17) and FLUX.1 (keys 14, 16) two-stream forwards and graph-replayed rollouts
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_round3a.sh'
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $OUT; rm -rf $OUT/*
MI355_NEXT=1 timeout 600 python -m pytest tests/test_gpu_next_round.py -q 2>&1 | tail -15 > $OUT/next_round_tests.log; echo "next rc=$?" >> $OUT/status
cat $OUT/next_round_tests.log
timeout 120 python scripts/clock_under_load.py --ms 30 > $OUT/clock_under_load.txt 2>&1; echo "clock rc=$?" >> $OUT/status
cat $OUT/clock_under_load.txt
# d = 64 attention variants incl. 3 = row sums on the matrix pipe (MSUM): parity vs fp32 and TFLOP/s at the bench shapes
timeout 120 python scripts/attn_ab.py > $OUT/attn_ab.txt 2>&1; echo "attn ab rc=$?" >> $OUT/status
cat $OUT/attn_ab.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/tests.log; echo "tests rc=$?" >> $OUT/status
cat $OUT/tests.log
# Qwen-Image two-stream forward (key 12) and hipGraph replay of the loop (key 17), both opt-in: A/B in one process (41 GB of synthetic weights: ~1 min to bind)
# (shapes: the bench shape B = 2 at 1024^2 and the reference's lora example shape B = 1 at 512^2)
timeout 500 python scripts/qwen_bench.py --batch 2 --denoise-steps 2 --iters 2 --ab-two-stream --ab-shapes 2x1024,1x512,1x384 2>&1 | grep '^{' > $OUT/qwen_two_stream_ab.txt; echo "qwen ab rc=$?" >> $OUT/status
cat $OUT/qwen_two_stream_ab.txt
# FLUX.1 double blocks two-stream (key 14) and hipGraph replay of the loop (key 16), both opt-in: the reference's example shapes (B = 1 at 384^2, B = 2 at 512^2) and the bench shapes
timeout 400 python scripts/flux_bench.py --denoise-steps 4 --iters 2 --ab-two-stream 1x384,2x512,1x1024,2x1024,8x1024 2>&1 | grep '^{' > $OUT/flux_two_stream_ab.txt; echo "flux ab rc=$?" >> $OUT/status
cat $OUT/flux_two_stream_ab.txt
# Wan2.1-1.3B: hipGraph replay of the loop (key 18) at the reference's example shape (240 x 240 x 5 frames, B = 1, CFG 5) and a mid-size clip
timeout 200 python scripts/wan_bench.py --batch 1 --height 240 --width 240 --frames 5 --denoise-steps 10 --iters 3 --ab-graph 2>&1 | grep '^{' > $OUT/wan_graph_ab.txt
timeout 200 python scripts/wan_bench.py --batch 1 --height 480 --width 832 --frames 17 --denoise-steps 4 --iters 2 --ab-graph 2>&1 | grep '^{' >> $OUT/wan_graph_ab.txt; echo "wan ab rc=$?" >> $OUT/status
cat $OUT/wan_graph_ab.txt
