#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/flux_bench.py --attn-only 2>&1 | tee gpurun_out/flux_attn_ab.log
