#!/bin/bash
# Round 6, call 20: the third-stream knob (key 11: V^T and the dual attention's projections beside the joint attention) at the B = 2 shapes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06t; mkdir -p $O
LEAN="--no-cpu-baseline --no-small-batch --no-vae --no-clock-probe --no-families --no-train-step --no-kernel-timing --no-selfcheck"
for i in 1 2; do for t in "11=0" "11=1"; do
  MI355_TUNE="$t" timeout 300 python bench.py $LEAN --batch 2 --size 1024 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b2_1024 tune=$t', d['value'], d['ms_per_step'])" >> $O/three_stream.txt
  MI355_TUNE="$t" timeout 300 python bench.py $LEAN --batch 2 --size 512 --guidance 4.5 --denoise-steps 10 --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b2_512_cfg tune=$t', d['value'], d['ms_per_step'])" >> $O/three_stream.txt
done; done
cat $O/three_stream.txt
