#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/wan_bench.py --batch 2 --denoise-steps 2 2>&1 | tail -2 | tee gpurun_out/wan_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wan_prof -- python $GRAFT_REPO_ROOT/scripts/wan_bench.py --batch 2 --denoise-steps 1 --iters 1 > $GRAFT_REPO_ROOT/gpurun_out/wan_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py gpurun_out/wan_prof > gpurun_out/wan_prof_summary.txt 2>&1 || true
find gpurun_out/wan_prof -size +1M -delete
head -22 gpurun_out/wan_prof_summary.txt
