#!/bin/bash
# round 4, closing call: the -m gpu suite at HEAD minus the host-core-heavy full-size file (run in full at the evidence commit b5da294) and the bench-leg test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04r; mkdir -p $O
echo "commit ${MI355_COMMIT:-unknown}" > $O/pytest_gpu_head.txt
( time timeout 460 python -m pytest tests -q -m gpu --ignore=tests/test_gpu_fullsize.py --deselect tests/test_gpu_schedules.py::test_bench_small_batch_legs_report_numbers ) >> $O/pytest_gpu_head.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_head.txt
tail -n 8 $O/pytest_gpu_head.txt
