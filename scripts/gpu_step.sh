#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_flux.py tests/test_gpu_wan.py -q -k "stepwise" 2>&1 | tail -25
