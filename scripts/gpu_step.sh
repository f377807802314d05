#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/attn_ab.py > gpurun_out/attn_ab.log 2>&1; cat gpurun_out/attn_ab.log
