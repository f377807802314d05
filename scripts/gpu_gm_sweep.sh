set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gm
F="--no-cpu-baseline --no-selfcheck --no-vae --steps 2 --warmup 1 --no-kernel-timing"
for GM in 6 0 4 8 12 6; do
  echo -n "gm=$GM " >> gpurun_out/gm/sweep.log
  MI355_TUNE="7=$GM" timeout 300 python bench.py $F 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/gm/sweep.log
done
cat gpurun_out/gm/sweep.log
