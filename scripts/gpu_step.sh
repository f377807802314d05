#!/bin/bash
# conv A/B + effect of the cheaper GELU on the rollout bench + regression tests of the touched kernels
mkdir -p gpurun_out
timeout 600 python scripts/conv_ab.py > gpurun_out/conv_ab.log 2>&1; cat gpurun_out/conv_ab.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q > gpurun_out/kernel_tests.log 2>&1; tail -3 gpurun_out/kernel_tests.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_gelu.log 2>&1; tail -2 gpurun_out/bench_gelu.log
