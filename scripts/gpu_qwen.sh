set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/qwen
timeout 900 python -m pytest tests/test_gpu_qwen.py tests/test_gpu_flux.py tests/test_gpu_wan.py -q -s 2>&1 | tail -40 | tee gpurun_out/qwen/tests.log
timeout 600 python scripts/qwen_bench.py --batch 2 --denoise-steps 2 2>&1 | tail -3 | tee gpurun_out/qwen/bench.log
