# head_dim-64 forward attention: 2-stage vs 4-stage K / V^T ring (key 23), unit + invariance tests under ring 4, then the bench shape
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03x
mkdir -p $OUT
(MI355_TUNE="23=4" timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward.py -m gpu -q -x -k "attention or bit_identical" 2>&1 | tail -3) > $OUT/pytest_ring4.log
cat $OUT/pytest_ring4.log
for rep in 1 2; do
for cfg in "23=2" "23=4" "23=2,6=0" "23=4,6=0"; do
  tag=$(echo $cfg | tr -d '=,')
  MI355_TUNE=$cfg timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --no-selfcheck 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('$cfg', 'value', d['value'], 'attn frac', r['frac'], 'achieved', r['achieved'], 'fwd', r.get('forward',{}).get('frac'))
" >> $OUT/ring_ab.txt
done
done
cat $OUT/ring_ab.txt
