#!/bin/bash
# first GPU pass of the VAE decode: parity tests, microbench, per-kernel profile
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vae.py -x -q > gpurun_out/vae_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/vae_tests.log
tail -15 gpurun_out/vae_tests.log
timeout 600 python scripts/vae_bench.py --batch 2 --conv-cfg 0 2 4 > gpurun_out/vae_bench.log 2>&1
timeout 300 python scripts/vae_bench.py --batch 4 --conv-cfg 0 >> gpurun_out/vae_bench.log 2>&1
cat gpurun_out/vae_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vae_prof -- python $GRAFT_REPO_ROOT/scripts/vae_bench.py --batch 2 --iters 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py gpurun_out/vae_prof > gpurun_out/vae_prof_summary.txt 2>&1 || true
find gpurun_out/vae_prof -size +1M -delete
head -40 gpurun_out/vae_prof_summary.txt
