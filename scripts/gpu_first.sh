set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/hw.log; nproc >> gpurun_out/hw.log; cat gpurun_out/hw.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -60 > gpurun_out/t_kernels.log
cat gpurun_out/t_kernels.log
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q 2>&1 | tail -80 > gpurun_out/t_model.log
cat gpurun_out/t_model.log
timeout 300 python scripts/microbench.py > gpurun_out/microbench.log 2>&1
cat gpurun_out/microbench.log
