# in-model A/B of the head_dim-128 attention: default (4-wave hand-scheduled kernel) vs 8-wave kernels only (MI355_TUNE=5=5); then the family tests
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $OUT
for t in "5=5" "5=0"; do
  MI355_TUNE=$t timeout 500 python bench.py --model flux1 --steps 1 --warmup 1 --no-vae 2>/dev/null | tail -1 > $OUT/flux_$t.json
  MI355_TUNE=$t timeout 400 python scripts/wan_bench.py --batch 2 --denoise-steps 2 2>/dev/null | tail -1 > $OUT/wan_$t.json
  MI355_TUNE=$t timeout 400 python scripts/qwen_bench.py --batch 2 --denoise-steps 2 2>/dev/null | tail -1 > $OUT/qwen_$t.json
  python - <<PY
import json
for m in ("flux","wan","qwen"):
    d=json.loads(open("$OUT/%s_$t.json" % m).read())
    print("$t", m, d.get("value", d.get("denoise_steps_per_s")), d.get("roofline",{}).get("frac", d.get("frac_of_2.5PF")))
PY
done
timeout 900 python -m pytest tests/test_gpu_flux.py tests/test_gpu_wan.py tests/test_gpu_qwen.py tests/test_gpu_schedules.py -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
