# per-(kernel, grid) breakdown of one optimize() replay step (default targets) + the -s prints of the config D / E / band parity tests
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03s
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_train -o train -- python $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 2 --size 1024 --train attn --iters 3 --only-step > $OUT/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > $OUT/train_kernel_by_grid.txt 2>&1
import csv, glob, os, collections
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03s")
f = glob.glob(os.path.join(out, "prof_train", "**", "*kernel_trace*.csv"), recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last quarter of the launches = the last of the 4 profiled steps (steps are identical launch sequences)
names = [r["Kernel_Name"] for r in rows]
n = len(rows)
step = None
for cand in range(n // 5, n // 3 + 1):
    if names[n - cand:] == names[n - 2 * cand:n - cand]:
        step = cand
        break
print("launches total", n, "per step", step)
last = rows[n - step:] if step else rows
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
print("last step wall (first start to last end) ms", (t1 - t0) / 1e6)
for r in last:
    g = (r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    a = acc[g]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in acc.values())
print("sum of kernel durations ms", tot / 1e3)
for g, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{a[1]/1e3:8.2f} ms {100*a[1]/tot:5.1f}% calls {a[0]:5d} avg {a[1]/a[0]:8.1f} us  grid {g[1]}x{g[2]}x{g[3]} wg {g[4]}  {g[0]}")
PY
head -80 $OUT/train_kernel_by_grid.txt
(timeout 900 python -m pytest tests/test_gpu_wan.py tests/test_gpu_qwen.py tests/test_gpu_fullsize.py -m gpu -q -s -k "20280 or 1328 or bf16_band or independent_of_the_batch" 2>&1 | grep -v "^$" | tail -40) > $OUT/parity_prints.log
cat $OUT/parity_prints.log
find $OUT -type f -size +1M -delete
