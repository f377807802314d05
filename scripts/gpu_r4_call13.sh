#!/bin/bash
# round 4, thirteenth GPU call: generate the config-B oracle fixture on the box (GPU-drawn weights, host-core oracle), then the WHOLE -m gpu suite
# with per-test durations at this commit
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $O
( time timeout 1200 python oracle/make_config_b_golden.py $O/config_b_oracle.npz ) > $O/make_golden.log 2>&1
cp $O/config_b_oracle.npz tests/golden/config_b_oracle.npz
( time timeout 2400 python -m pytest tests -q -m gpu --durations=40 -s ) > $O/pytest_gpu_full.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_full.txt
find $O -type f -size +4M -delete
tail -n 4 $O/make_golden.log
grep -h "passed\|failed\|rc=\|^real\|Error" $O/pytest_gpu_full.txt | tail -n 8
grep -A42 "slowest 40 durations" $O/pytest_gpu_full.txt | head -n 44
