# Round-4 evidence set at ONE commit, shipped defaults.  MI355_COMMIT = git sha of the snapshot (the box has no .git).
#   tests    the whole -m gpu suite with per-test durations
#   bench    the driver's command shape (python bench.py: power, families, the three optimize() steps in the line) + all-classes + CFG variant
#   stats1   rocprofv3 --kernel-trace --stats, two-stream forward OFF (MI355_TUNE=8=0): clean per-kernel durations
#   stats2   the same with the shipped defaults (two-stream ON): wall clock under the profiler, overlapping kernels
#   pmc      FETCH_SIZE / WRITE_SIZE / MFMA-busy passes (separate runs, --pmc only; two-stream OFF so that counters belong to one kernel at a time)
# usage: gpurun --timeout 2400 -- 'MI355_COMMIT=<sha> bash scripts/gpu_r4_evidence.sh'      PARTS="tests bench stats1 stats2 pmc"
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4final
mkdir -p $OUT; rm -rf $OUT/prof_*
PARTS=${PARTS:-"bench tests stats1 pmc stats2"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_ROUND=4
echo "commit ${MI355_COMMIT:-unknown}" > $OUT/commit.txt
if has bench; then
  ( time timeout 900 python bench.py 2>$OUT/bench_default.err ) > $OUT/bench_b8_ncfg1.json 2> $OUT/bench_default.time; cut -c1-400 $OUT/bench_b8_ncfg1.json; tail -n 3 $OUT/bench_default.time
  timeout 600 python bench.py --no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --kernel-timing all 2>/dev/null > $OUT/bench_b8_ncfg1_allclasses.json
  timeout 600 python bench.py --no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --guidance 4.5 --batch 4 2>/dev/null > $OUT/bench_b4_ncfg2.json
fi
if has stats1; then
  (cd /tmp && MI355_TUNE="8=0" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_single -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-families --no-train-step > $OUT/prof_stats1.log 2>&1)
  python scripts/summarize_prof.py $OUT prof_stats_single > $OUT/prof_summary_single_stream.txt 2>&1; head -n 30 $OUT/prof_summary_single_stream.txt
fi
if has stats2; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_two -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-vae --no-families --no-train-step > $OUT/prof_stats2.log 2>&1)
  grep '^{' $OUT/prof_stats2.log > $OUT/bench_under_rocprof_two_stream.json
  python scripts/summarize_prof.py $OUT prof_stats_two > $OUT/prof_summary_two_stream.txt 2>&1; head -n 12 $OUT/prof_summary_two_stream.txt
fi
if has pmc; then
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $C | tr ' ' '_')
    (cd /tmp && MI355_TUNE="8=0" timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-timing --no-selfcheck --no-vae --no-small-batch --no-clock-probe --no-families --no-train-step > $OUT/prof_pmc_$tag.log 2>&1)
  done
  python scripts/summarize_prof.py $OUT prof_stats_single > $OUT/prof_summary_pmc.txt 2>&1; grep -A40 "== PMC" $OUT/prof_summary_pmc.txt | head -n 60
fi
if has tests; then
  ( time timeout 1800 python -m pytest tests -q -m gpu --durations=15 -s ) > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.txt
  grep -h "passed\|failed\|rc=\|^real" $OUT/pytest_gpu.txt | tail -n 5
fi
find $OUT -type f -size +1M -delete
ls $OUT
