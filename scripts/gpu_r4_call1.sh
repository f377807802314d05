#!/bin/bash
# round 4, first GPU call: (1) key 25 verified at full width against the gradient band, (2) the config-B / advantages / RCCL tests, (3) SMI probe
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04a; mkdir -p $O
timeout 120 python scripts/power_meter.py > $O/power_probe.txt 2>&1
(timeout 120 rocm-smi --showpower --showclocks 2>&1 | head -30) > $O/rocm_smi.txt
MI355_TUNE=25=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s -m gpu \
   -k "config_b_rollout or config_b_cfg or advantages or replay_gradients" > $O/pytest_fullsize_key25.txt 2>&1
echo "rc=$?" >> $O/pytest_fullsize_key25.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -s -m gpu -k "replay_gradients" > $O/pytest_grad_default.txt 2>&1
echo "rc=$?" >> $O/pytest_grad_default.txt
timeout 300 python -m pytest tests/test_gpu_ddp_rccl.py -x -q -s -m gpu > $O/pytest_ddp_rccl.txt 2>&1
echo "rc=$?" >> $O/pytest_ddp_rccl.txt
tail -5 $O/*.txt
