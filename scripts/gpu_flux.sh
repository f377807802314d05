#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_flux.py -x -q > gpurun_out/flux_tests.log 2>&1; echo "exit $?" >> gpurun_out/flux_tests.log; tail -25 gpurun_out/flux_tests.log
timeout 300 python scripts/flux_bench.py --attn-only > gpurun_out/flux_bench.log 2>&1
timeout 900 python scripts/flux_bench.py --batch 8 --denoise-steps 2 --iters 1 >> gpurun_out/flux_bench.log 2>&1
cat gpurun_out/flux_bench.log
