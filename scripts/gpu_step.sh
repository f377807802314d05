#!/bin/bash
# scratch GPU pass (edited per experiment): full GPU suite + smoke
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tee gpurun_out/final_smoke.log
