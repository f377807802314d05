set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/small
F="--no-cpu-baseline --no-selfcheck --no-vae --steps 2 --warmup 1"
for V in 1 2; do
  for CFG in "--batch 2 --size 1024" "--batch 1 --size 1024" "--batch 4 --size 1024" "--batch 2 --size 512 --guidance 4.5 --denoise-steps 10" "--batch 8 --size 512 --denoise-steps 10" "--batch 8 --size 1024"; do
    echo "variant $V $CFG" >> gpurun_out/small/ab.log
    MI355_TUNE="1=$V" timeout 300 python bench.py $F $CFG 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['forward']['frac'], d['roofline'].get('ms_per_launch'))" >> gpurun_out/small/ab.log
  done
done
cat gpurun_out/small/ab.log
timeout 600 python scripts/qwen_bench.py --batch 1 --denoise-steps 2 --size 1328 --dynamics ODE 2>&1 | tail -1 | tee gpurun_out/small/qwen1328.log
