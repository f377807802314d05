#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/flux_bench.py --attn-only > gpurun_out/flux_attn_ab.log 2>&1; cat gpurun_out/flux_attn_ab.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/flux_prof -- python $GRAFT_REPO_ROOT/scripts/flux_bench.py --batch 8 --denoise-steps 2 --iters 1 > $GRAFT_REPO_ROOT/gpurun_out/flux_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py gpurun_out/flux_prof > gpurun_out/flux_prof_summary.txt 2>&1 || true
find gpurun_out/flux_prof -size +1M -delete
head -32 gpurun_out/flux_prof_summary.txt; tail -2 gpurun_out/flux_prof.log
timeout 900 python bench.py --model flux1 --steps 1 --warmup 1 --denoise-steps 8 > gpurun_out/bench_flux.log 2>&1; tail -1 gpurun_out/bench_flux.log
