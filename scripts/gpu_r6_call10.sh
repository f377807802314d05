#!/bin/bash
# Round 6, call 10: GroupNorm statistics in the convolution epilogue (key 40): VAE tests, decode A/B, per-kernel table of the decode.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06j; mkdir -p $O
( time timeout 900 python -m pytest -q -s -m gpu tests/test_gpu_vae.py tests/test_gpu_wan_vae.py "tests/test_gpu_fullsize.py::test_config_a_advantages_from_engine_images_match_the_oracle_pipeline" ) > $O/pytest_vae.txt 2>&1; echo "rc=$?" >> $O/pytest_vae.txt
grep -h "passed\|failed\|rc=\|Error\|real\|FAILED\|VAE decode" $O/pytest_vae.txt | cut -c1-300 | tail -n 12
for t in "40=1" "40=0" "40=1" "40=0"; do
  MI355_TUNE="$t" timeout 300 python scripts/vae_bench.py --batch 4 --iters 5 2>/dev/null | tail -n 1 | sed "s/^/tune=$t /" >> $O/vae_ab.txt
done
cat $O/vae_ab.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vae -o vae -- python $GRAFT_REPO_ROOT/scripts/vae_bench.py --batch 4 --iters 3 > $O/prof_vae.log 2>&1)
python scripts/summarize_prof.py $O prof_vae > $O/vae_kernel_stats.txt 2>&1; find $O -type f -size +2M -delete; head -n 24 $O/vae_kernel_stats.txt | cut -c1-120
