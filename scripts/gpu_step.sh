#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flux.py -q -k "attention" 2>&1 | tail -3
timeout 300 python scripts/attn_ab.py 2>&1 | grep "variant 1" | head -4
