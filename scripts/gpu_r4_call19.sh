#!/bin/bash
# round 4, the last GPU seconds: (1) the shared kernels edited for the Wan backward (attention128_bwd S_kv fields, norm_rope_full rstd output) still pass
# the FLUX.1 backward and Wan forward suites; (2) first contact of the Wan native backward's tiny tests (opt-in), whatever they say
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04s; mkdir -p $O
timeout 110 python -m pytest tests/test_gpu_flux_backward.py tests/test_gpu_wan.py -q -m gpu -x > $O/pytest_shared_kernels.txt 2>&1; echo "rc=$?" >> $O/pytest_shared_kernels.txt
MI355_WAN_NATIVE_BACKWARD=1 timeout 60 python -m pytest tests/test_gpu_wan_backward.py -q -s -m gpu -k "norm_rope_full or replay_gradients" > $O/pytest_wan_backward_first_contact.txt 2>&1; echo "rc=$?" >> $O/pytest_wan_backward_first_contact.txt
tail -n 4 $O/pytest_shared_kernels.txt; grep -h "passed\|failed\|rel-L2\|Error\|rc=" $O/pytest_wan_backward_first_contact.txt | cut -c1-300 | tail -n 14
