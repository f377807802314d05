#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/attn_ab.py > gpurun_out/attn_ab.log 2>&1; cat gpurun_out/attn_ab.log
timeout 600 python -m pytest tests/test_gpu_vae.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_step.log 2>&1; tail -1 gpurun_out/bench_step.log
