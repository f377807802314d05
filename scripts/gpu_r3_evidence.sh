# Round-3 evidence set at ONE commit, shipped defaults (VERDICT r2 "Next round" item 2).  MI355_COMMIT = git sha of the snapshot.
#   stats1   rocprofv3 --kernel-trace --stats, two-stream forward OFF (MI355_TUNE=8=0): clean per-kernel durations
#   stats2   the same with the shipped defaults (two-stream ON): wall clock under the profiler, overlapping kernels
#   pmc      FETCH_SIZE / WRITE_SIZE / MFMA-busy passes (separate runs, two-stream OFF so that counters belong to one kernel at a time)
#   bench    the driver's command (python bench.py) + all-classes + CFG variant
# usage: gpurun --timeout 2400 -- 'MI355_COMMIT=<sha> bash scripts/gpu_r3_evidence.sh'      PARTS="bench stats1 stats2 pmc extra"
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $OUT; rm -rf $OUT/prof_* 
PARTS=${PARTS:-"bench stats1 stats2 pmc extra"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
export TMPDIR=/tmp
if has bench; then
  timeout 900 python bench.py 2>/dev/null > $OUT/bench_b8_ncfg1.json; cut -c1-600 $OUT/bench_b8_ncfg1.json
  timeout 600 python bench.py --no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --kernel-timing all 2>/dev/null > $OUT/bench_b8_ncfg1_allclasses.json
  timeout 600 python bench.py --no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --guidance 4.5 --batch 4 2>/dev/null > $OUT/bench_b4_ncfg2.json
fi
if has stats1; then
  (cd /tmp && MI355_TUNE="8=0" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe > $OUT/prof_stats.log 2>&1)
  python scripts/summarize_prof.py $OUT > $OUT/prof_summary_single_stream.txt 2>&1; head -30 $OUT/prof_summary_single_stream.txt
  mv $OUT/prof_stats $OUT/prof_stats_single
fi
if has stats2; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-vae > $OUT/prof_stats2.log 2>&1)
  grep '^{' $OUT/prof_stats2.log > $OUT/bench_under_rocprof_two_stream.json
  python scripts/summarize_prof.py $OUT > $OUT/prof_summary_two_stream.txt 2>&1; head -12 $OUT/prof_summary_two_stream.txt
  mv $OUT/prof_stats $OUT/prof_stats_two
fi
if has pmc; then
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $C | tr ' ' '_')
    (cd /tmp && MI355_TUNE="8=0" timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-timing --no-selfcheck --no-vae --no-small-batch --no-clock-probe > $OUT/prof_pmc_$tag.log 2>&1)
  done
  python scripts/summarize_prof.py $OUT > $OUT/prof_summary_pmc.txt 2>&1; grep -A40 "== PMC" $OUT/prof_summary_pmc.txt | head -60
fi
if has extra; then
  timeout 900 python bench.py --model flux1 --steps 1 --warmup 1 2>/dev/null > $OUT/bench_flux.json
  timeout 600 python scripts/wan_bench.py --batch 2 --denoise-steps 2 2>/dev/null | tail -1 > $OUT/bench_wan.json
  timeout 600 python scripts/qwen_bench.py --batch 2 --denoise-steps 2 2>/dev/null | tail -1 > $OUT/bench_qwen.json
  timeout 600 python scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 2 2>/dev/null | tail -1 > $OUT/bench_train.json
fi
find $OUT -type f -size +1M -delete
ls $OUT
