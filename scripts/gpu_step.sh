#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/attn_ab.py 2>&1 | tee gpurun_out/attn_ab.log
