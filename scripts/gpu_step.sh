#!/bin/bash
# clean per-kernel table of the shipped kernels (no self-check variants) + the reference's own example shapes (512^2)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; rm -rf $OUT/prof_stats
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck > $OUT/prof_stats.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1
find $OUT -type f -size +1M -delete
head -30 $OUT/prof_summary.txt
for cfg in "--size 512 --batch 8 --denoise-steps 10" "--size 512 --batch 8 --guidance 4.5 --denoise-steps 10" "--size 512 --batch 2 --guidance 4.5 --denoise-steps 10" "--batch 16"; do
  echo "== $cfg"
  timeout 600 python bench.py $cfg --no-cpu-baseline --no-vae --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'attn', r['achieved'], 'fwd', r['forward']['achieved'], r['forward']['frac'])"
done 2>&1 | tee $OUT/bench_shapes.log
