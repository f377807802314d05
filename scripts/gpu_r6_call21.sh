#!/bin/bash
# Round 6, call 21: the hardened `optimize_step_ddp` leg (rank agreement before the first DDP collective, deadline) on a world-size-1 RCCL group.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
O=$GRAFT_REPO_ROOT/gpurun_out/r06u; mkdir -p $O
( time timeout 600 python bench.py --ddp-step-world1 --no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --no-families --steps 1 --warmup 1 2>$O/err.txt ) > $O/bench_ddp_world1.json 2> $O/time.txt
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06u/bench_ddp_world1.json") if l.startswith("{")][-1])
print(d["value"], json.dumps(d.get("optimize_step_ddp"))[:500])
PY
tail -n 3 $O/time.txt; tail -n 3 $O/err.txt | cut -c1-300
