set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wvae
timeout 900 python -m pytest tests/test_gpu_wan_vae.py tests/test_gpu_vae.py -q -s -x 2>&1 | tail -40 | tee gpurun_out/wvae/tests.log
timeout 600 python scripts/wan_vae_bench.py 2>&1 | tail -4 | tee gpurun_out/wvae/bench.log
