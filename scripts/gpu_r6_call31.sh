#!/bin/bash
# Round 6, call 31: per-kernel tables of the Wan2.1 / FLUX.1 optimize() steps with the pipelined head_dim-128 attention backward.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab; mkdir -p $O
sed -e 's#gpurun_out/r06v#gpurun_out/r06ab#g' -e 's#O=$GRAFT_REPO_ROOT/gpurun_out/r06v#O=$GRAFT_REPO_ROOT/gpurun_out/r06ab#' scripts/gpu_r6_call22.sh > /tmp/c22.sh
bash /tmp/c22.sh
