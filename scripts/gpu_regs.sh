set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/regs
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_backward.py tests/test_gpu_grpo_epoch.py -q -x 2>&1 | tail -4 | tee gpurun_out/regs/tests.log
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/regs/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/regs/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print("value", d['value'], "attn frac", r['frac'], "forward", r['forward']['frac'], "dynamic", r.get('attention_dynamic'), "by_class", json.dumps(r.get('by_class'))[:400])
PY
timeout 600 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 2 2>/dev/null | tail -1 | tee gpurun_out/regs/train.log
timeout 300 python scripts/gemm_trace.py 2>&1 | tail -6 | tee gpurun_out/regs/trace.log
