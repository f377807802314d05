#!/bin/bash
# Round 6, call 28: the launch-schedule tests at the final schedule (row-major weight gradients for the text rows and for Wan), traces re-recorded.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06y; mkdir -p $O $O/traces
( time MI355_DUMP_TRACES=$O/traces timeout 1200 python -m pytest tests/test_gpu_schedules.py -q -m gpu ) > $O/pytest_schedules.txt 2>&1; echo "rc=$?" >> $O/pytest_schedules.txt
grep -h "passed\|failed\|rc=\|real\|FAILED\|Error\|assert" $O/pytest_schedules.txt | cut -c1-300 | tail -n 12
ls -la $O/traces | head -20
