#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_adapter.py tests/test_gpu_rollout_variants.py -q 2>&1 | tail -4
for v in 1 0 1 0; do
MI355_TWO_STREAM=$v timeout 600 python bench.py --no-cpu-baseline --no-vae --no-selfcheck --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('two_stream=$v', d['value'], d['ms_per_step'], 'attn', r['achieved'], 'fwd', r['forward']['achieved'], r['forward']['frac'])"
done
