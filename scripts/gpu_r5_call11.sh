#!/bin/bash
# Round 5, call 11 (last): the w4 minimum-grid rule A/B (key 31: 512 = new default, 0 = the old dispatch), then the whole -m gpu suite, smoke() and the
# bench line at the final HEAD.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_ROUND=5
O=$GRAFT_REPO_ROOT/gpurun_out/r05k; mkdir -p $O
echo "commit ${MI355_COMMIT:-unknown}" > $O/commit.txt
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --no-kernel-timing"
one() { MI355_TUNE="$2" timeout 200 python bench.py $3 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 tune=[$2]', d['value'])" >> $O/w4_min_tiles_ab.txt; }
for t in "31=0" "" "31=0" ""; do one b2_512_cfg "$t" "--steps 30 --warmup 4 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10"; done
for t in "31=0" "" ; do one b4_512 "$t" "--steps 30 --warmup 4 --size 512 --batch 4 --denoise-steps 10"; done
for t in "31=0" "" ; do one b2_1024 "$t" "--steps 6 --warmup 2 --size 1024 --batch 2"; done
for t in "31=0" "" ; do MI355_TUNE="$t" timeout 200 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 5 --only-step 2>/dev/null | sed "s/^/train_b2_1024 tune=[$t] /" >> $O/w4_min_tiles_ab.txt; done
cat $O/w4_min_tiles_ab.txt
( time timeout 1800 python -m pytest tests -q -m gpu --durations=8 ) > $O/pytest_gpu_full.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_full.txt
grep -h "passed\|failed\|rc=\|^real\|FAILED" $O/pytest_gpu_full.txt | tail -n 8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -n 2 > $O/smoke.txt; cat $O/smoke.txt
( time timeout 900 python bench.py 2>/dev/null ) > $O/bench_b8_ncfg1.json 2> $O/bench_default.time; cut -c1-260 $O/bench_b8_ncfg1.json; tail -n 3 $O/bench_default.time
