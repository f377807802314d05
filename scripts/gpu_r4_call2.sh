#!/bin/bash
# round 4, second GPU call: first run of the native FLUX.1 backward + the fixed RCCL DDP test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flux_backward.py -q -s -m gpu > $O/pytest_flux_backward.txt 2>&1
echo "rc=$?" >> $O/pytest_flux_backward.txt
timeout 300 python -m pytest tests/test_gpu_ddp_rccl.py -x -q -s -m gpu > $O/pytest_ddp_rccl.txt 2>&1
echo "rc=$?" >> $O/pytest_ddp_rccl.txt
timeout 600 python -m pytest tests/test_gpu_flux.py -x -q -m gpu > $O/pytest_flux_forward.txt 2>&1
echo "rc=$?" >> $O/pytest_flux_forward.txt
tail -n 5 $O/*.txt
