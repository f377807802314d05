# First GPU calls of round 3: everything here was written after round 2's GPU budget was spent and has NOT run on an MI355X yet.
#   tests  the gated checks (tests/test_gpu_next_round.py): SD3.5-large width forward vs the oracle; bench.py's small-batch legs; two-stream /
#          graph-replay bit-identity for FLUX.1, Qwen-Image, Wan; attention variant 3                                     (~4 min)
#   clock  the delivered-clock probe beside the hot kernels (scripts/clock_under_load.py): which kernel has headroom at ITS clock (~1 min)
#   attn   d = 64 attention variants incl. 3 = row sums on the matrix pipe: parity vs fp32 and TFLOP/s at the bench shapes  (~0.5 min)
#   suite  the whole -m gpu suite at HEAD                                                                                  (~5 min)
#   qwen   A/B of the Qwen-Image two-stream forward (key 12) x hipGraph replay of the loop (key 17), one process            (~4 min)
#   flux   the same for FLUX.1 (keys 14, 16) over the reference's example shapes and the bench shapes                       (~3 min)
#   wan    Wan graph replay (key 18) at the reference's example shape and a mid-size clip                                   (~1 min)
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round3a.sh'                 (all parts)
#        gpurun --timeout 600  -- 'PARTS="tests clock attn" bash scripts/gpu_round3a.sh'
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $OUT
PARTS=${PARTS:-"tests clock attn suite qwen flux wan"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }

if has tests; then
  MI355_NEXT=1 timeout 600 python -m pytest tests/test_gpu_next_round.py -q > $OUT/next_round_tests.log 2>&1; echo "next rc=$?" >> $OUT/status
  tail -25 $OUT/next_round_tests.log
fi
if has clock; then
  timeout 120 python scripts/clock_under_load.py --ms 30 > $OUT/clock_under_load.txt 2>&1; echo "clock rc=$?" >> $OUT/status
  cat $OUT/clock_under_load.txt
fi
if has attn; then
  timeout 120 python scripts/attn_ab.py > $OUT/attn_ab.txt 2>&1; echo "attn ab rc=$?" >> $OUT/status
  cat $OUT/attn_ab.txt
fi
if has suite; then
  timeout 700 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/status
  tail -8 $OUT/tests.log
fi
if has qwen; then   # 41 GB of synthetic weights: ~1 min to bind; shapes: the bench shape and the reference's example shapes (B = 1 at 512^2 / 384^2)
  timeout 500 python scripts/qwen_bench.py --batch 2 --denoise-steps 2 --iters 2 --ab-two-stream --ab-shapes 2x1024,1x512,1x384 > $OUT/qwen_ab.log 2>&1; echo "qwen ab rc=$?" >> $OUT/status
  grep '^{' $OUT/qwen_ab.log > $OUT/qwen_two_stream_ab.txt; cat $OUT/qwen_two_stream_ab.txt; tail -3 $OUT/qwen_ab.log
fi
if has flux; then   # the reference's example shapes (B = 1 at 384^2, B = 2 at 512^2) and the bench shapes
  timeout 400 python scripts/flux_bench.py --denoise-steps 4 --iters 2 --ab-two-stream 1x384,2x512,1x1024,2x1024,8x1024 > $OUT/flux_ab.log 2>&1; echo "flux ab rc=$?" >> $OUT/status
  grep '^{' $OUT/flux_ab.log > $OUT/flux_two_stream_ab.txt; cat $OUT/flux_two_stream_ab.txt; tail -3 $OUT/flux_ab.log
fi
if has wan; then    # the reference's example shape (240 x 240 x 5 frames, B = 1, CFG 5) and a mid-size clip
  timeout 200 python scripts/wan_bench.py --batch 1 --height 240 --width 240 --frames 5 --denoise-steps 10 --iters 3 --ab-graph > $OUT/wan_ab.log 2>&1; echo "wan ab small rc=$?" >> $OUT/status
  timeout 200 python scripts/wan_bench.py --batch 1 --height 480 --width 832 --frames 17 --denoise-steps 4 --iters 2 --ab-graph >> $OUT/wan_ab.log 2>&1; echo "wan ab mid rc=$?" >> $OUT/status
  grep '^{' $OUT/wan_ab.log > $OUT/wan_graph_ab.txt; cat $OUT/wan_graph_ab.txt; tail -3 $OUT/wan_ab.log
fi
cat $OUT/status
