# Round-end evidence: GPU tests, smoke, bench lines, rocprofv3 stats + PMC of the bench command.  MI355_COMMIT = git sha of the snapshot.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $OUT; rm -rf $OUT/prof_* $OUT/final_*
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/final_tests.log; cat $OUT/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu > $OUT/final_smoke.log; cat $OUT/final_smoke.log
timeout 900 python bench.py 2>/dev/null > $OUT/final_bench.json; cut -c1-400 $OUT/final_bench.json
timeout 600 python bench.py --no-cpu-baseline --kernel-timing all 2>/dev/null > $OUT/final_bench_allclasses.json
timeout 600 python bench.py --no-cpu-baseline --guidance 4.5 --batch 4 2>/dev/null > $OUT/final_bench_cfg.json
timeout 900 python bench.py --model flux1 --steps 1 --warmup 1 2>/dev/null > $OUT/final_bench_flux.json
timeout 600 python scripts/wan_bench.py --batch 2 --denoise-steps 2 2>/dev/null | tail -1 > $OUT/final_bench_wan.json
timeout 600 python scripts/qwen_bench.py --batch 2 --denoise-steps 2 2>/dev/null | tail -1 > $OUT/final_bench_qwen.json
timeout 600 python scripts/wan_vae_bench.py 2>/dev/null | tail -2 > $OUT/final_bench_wan_vae.json
timeout 600 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 2 2>/dev/null | tail -1 > $OUT/final_bench_train.json
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck > $OUT/prof_stats.log 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-timing --no-selfcheck --no-vae > $OUT/prof_pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1

head -30 $OUT/prof_summary.txt
find $OUT -type f -size +1M -delete
