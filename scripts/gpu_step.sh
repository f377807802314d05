#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wan.py -x -q 2>&1 | tail -30
