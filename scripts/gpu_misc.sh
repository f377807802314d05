set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/misc
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_grpo_epoch.py -q -x 2>&1 | tail -5 | tee gpurun_out/misc/tests.log
timeout 600 python scripts/train_bench.py --batch 2 --size 1024 --train blocks --iters 2 2>&1 | tail -1 | tee gpurun_out/misc/train.log
timeout 600 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 2 2>&1 | tail -1 | tee -a gpurun_out/misc/train.log
timeout 600 python scripts/qwen_bench.py --batch 1 --denoise-steps 2 --size 1328 --dynamics ODE 2>&1 | tail -1 | tee gpurun_out/misc/qwen1328.log
