#!/bin/bash
# Round 6, call 6: gradient-noise isolation by tensor class; the ONE energy experiment (4-wave register tiles for the N = 1536 GEMMs, J per denoise
# step); PMC at the weak shapes with the mid-size kernel off and on (top-40 kernels per counter group).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O
( time timeout 600 python -m pytest -q -s -m gpu tests/test_gpu_fullsize.py -k "gradient_noise" ) > $O/pytest_gradient_noise.txt 2>&1; echo "rc=$?" >> $O/pytest_gradient_noise.txt
grep -h "gradient noise\|top ratio\|passed\|failed\|rc=\|Error" $O/pytest_gradient_noise.txt | cut -c1-330
bash scripts/gpu_r6_energy.sh
MID=0 bash scripts/gpu_r6_pmc.sh > $O/pmc_run_mid0.log 2>&1; grep "rc=" $O/pmc_run_mid0.log | tr '\n' ' '
MID=1 bash scripts/gpu_r6_pmc.sh > $O/pmc_run_mid1.log 2>&1; grep "rc=" $O/pmc_run_mid1.log | tr '\n' ' '
