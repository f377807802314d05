# Round-6 evidence set at ONE commit, shipped defaults.  MI355_COMMIT = git sha of the snapshot (the box has no .git).
#   bench    the driver's command shape (python bench.py, every leg) + all-classes + CFG variant
#   stats1   rocprofv3 --kernel-trace --stats, two-stream forward OFF (MI355_TUNE=8=0): clean per-kernel durations (bench.py keeps the mode it started with)
#   stats2   the same with the shipped defaults (two-stream ON)
#   pmc      FETCH_SIZE / WRITE_SIZE / MFMA-busy passes (separate runs, --pmc only)
#   train    rocprofv3 --kernel-trace --stats of the SD3.5 optimize() step on the reference's default target set; the same at config D (Wan)
#   small    per-kernel table of the reference's 512^2 B = 2 CFG example shape (single stream)
#   tests    PARTS contains "tests": the whole -m gpu suite;  "newtests": only what changed since the last full run
# usage: gpurun --timeout 2400 -- 'MI355_COMMIT=<sha> PARTS="bench stats1 ..." bash scripts/gpu_r6_evidence.sh'
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6last
mkdir -p $OUT; rm -rf $OUT/prof_*
PARTS=${PARTS:-"newtests bench stats1 pmc stats2 train small"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_ROUND=6
echo "commit ${MI355_COMMIT:-unknown}" > $OUT/commit.txt
LEAN="--no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step"
if has newtests; then
  ( time timeout 900 python -m pytest tests/test_gpu_wan_backward.py tests/test_gpu_flux_backward.py tests/test_gpu_fullsize.py -q -s -m gpu -k "one_block or one_double or replay_gradients_vs_oracle" ) > $OUT/pytest_newtests.txt 2>&1; echo "rc=$?" >> $OUT/pytest_newtests.txt
  grep -h "passed\|failed\|rc=\|best-fit\|Error" $OUT/pytest_newtests.txt | cut -c1-600 | tail -n 12
fi
if has bench; then
  ( time timeout 1200 python bench.py 2>$OUT/bench_default.err ) > $OUT/bench_b8_ncfg1.json 2> $OUT/bench_default.time; cut -c1-400 $OUT/bench_b8_ncfg1.json; tail -n 3 $OUT/bench_default.time
  timeout 600 python bench.py $LEAN --kernel-timing all 2>/dev/null > $OUT/bench_b8_ncfg1_allclasses.json
  timeout 600 python bench.py $LEAN --guidance 4.5 --batch 4 2>/dev/null > $OUT/bench_b4_ncfg2.json
fi
if has stats1; then
  (cd /tmp && MI355_TUNE="8=0" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_single -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-selfcheck $LEAN > $OUT/prof_stats1.log 2>&1)
  python scripts/summarize_prof.py $OUT prof_stats_single > $OUT/prof_summary_single_stream.txt 2>&1; head -n 18 $OUT/prof_summary_single_stream.txt
fi
if has stats2; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_two -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-selfcheck $LEAN > $OUT/prof_stats2.log 2>&1)
  grep '^{' $OUT/prof_stats2.log > $OUT/bench_under_rocprof_two_stream.json
  python scripts/summarize_prof.py $OUT prof_stats_two > $OUT/prof_summary_two_stream.txt 2>&1; head -n 10 $OUT/prof_summary_two_stream.txt
fi
if has pmc; then
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $C | tr ' ' '_')
    (cd /tmp && MI355_TUNE="8=0" timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-kernel-timing --no-selfcheck $LEAN > $OUT/prof_pmc_$tag.log 2>&1)
  done
  python scripts/summarize_prof.py $OUT prof_stats_single > $OUT/prof_summary_pmc.txt 2>&1; grep -A40 "== PMC" $OUT/prof_summary_pmc.txt | head -n 50
fi
if has train; then
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_sd3 -o train -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train default --iters 3 --only-step > $OUT/prof_train_sd3.log 2>&1)
  python - <<'PY' > $OUT/sd3_train_step_default_set_kernel_stats.txt 2>&1
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r6last/prof_train_sd3/**/*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f} {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
PY
  grep '^{' $OUT/prof_train_sd3.log >> $OUT/sd3_train_step_default_set_kernel_stats.txt; head -n 12 $OUT/sd3_train_step_default_set_kernel_stats.txt | cut -c1-200
fi
if has small; then
  (cd /tmp && MI355_TUNE="8=0" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_small -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10 --no-selfcheck --no-kernel-timing $LEAN > $OUT/prof_small.log 2>&1)
  python scripts/summarize_prof.py $OUT prof_stats_small > $OUT/prof_summary_512_b2_cfg_single_stream.txt 2>&1; head -n 18 $OUT/prof_summary_512_b2_cfg_single_stream.txt
fi
if has tests; then
  ( time timeout 1800 python -m pytest tests -q -m gpu --durations=15 -s ) > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.txt
  grep -h "passed\|failed\|rc=\|^real" $OUT/pytest_gpu.txt | tail -n 5
fi
find $OUT -type f -size +1M -delete
ls $OUT
