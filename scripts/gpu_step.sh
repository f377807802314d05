#!/bin/bash
mkdir -p gpurun_out
timeout 120 ./scripts/mb/mfma_valu_mix 2>&1 | tee gpurun_out/mfma_valu_mix.log
