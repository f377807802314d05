#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_flux.py -q -k "full_width" 2>&1 | tail -5
