#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q 2>&1 | tail -3
for cfg in "" "--batch 2" "--size 512 --batch 8 --denoise-steps 10"; do
  echo "== $cfg"
  timeout 600 python bench.py $cfg --no-cpu-baseline --no-vae --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'attn', r['achieved'], 'fwd', r['forward']['achieved'], r['forward']['frac'])"
done 2>&1 | tee gpurun_out/bench_dispatch.log
timeout 600 python bench.py --model flux1 --no-vae --steps 1 --warmup 1 --denoise-steps 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flux', d['value'], d['roofline']['achieved'])"
