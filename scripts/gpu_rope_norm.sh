# FLUX.1 / Qwen-Image q|k producer rewritten with 16-byte accesses: family tests, then the two rollout benches
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ac
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_flux.py tests/test_gpu_qwen.py tests/test_gpu_schedules.py -m gpu -q -x -k "not training_step and not sd3" 2>&1 | tail -3) > $OUT/pytest.log; cat $OUT/pytest.log
timeout 300 python bench.py --model flux1 --steps 1 --warmup 1 --no-vae 2>/dev/null | tail -1 > $OUT/bench_flux.json; python -c "
import json; d=json.loads(open('$OUT/bench_flux.json').readline()); print('flux', d['value'], d['roofline'].get('forward'))"
timeout 300 python scripts/qwen_bench.py --batch 2 --denoise-steps 2 2>/dev/null | tail -1 > $OUT/bench_qwen.json; cat $OUT/bench_qwen.json
