set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -60 > gpurun_out/t_gpu.log
cat gpurun_out/t_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; cat gpurun_out/smoke.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
timeout 600 python bench.py --steps 2 --warmup 1 --guidance 4.5 --batch 4 --no-cpu-baseline > gpurun_out/bench_cfg.log 2>> gpurun_out/bench.err; cat gpurun_out/bench_cfg.log
