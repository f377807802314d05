# Wan: data-dependent static / running-max split of the self-attention (key 24): tests, then the config-D bench with it on / off
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03aa
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_wan.py -m gpu -q -x -k "not 20280 and not 4608" 2>&1 | tail -3) > $OUT/pytest_wan.log
cat $OUT/pytest_wan.log
for rep in 1 2; do
for cfg in "24=1" "24=0"; do
  MI355_TUNE=$cfg timeout 300 python scripts/wan_bench.py --batch 2 --denoise-steps 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg', d['denoise_steps_per_s'], d['frac_of_2.5PF'])" >> $OUT/wan_ab.txt
done
done
cat $OUT/wan_ab.txt
