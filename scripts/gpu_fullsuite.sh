# the driver's round-end GPU tier, run by hand: full `pytest -m gpu`, smoke(), the default bench line
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3suite
mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $OUT/smoke.log
cat $OUT/smoke.log
