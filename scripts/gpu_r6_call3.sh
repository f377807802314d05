#!/bin/bash
# Round 6, call 3: the mid-size kernel with per-wave staggered LDS-DMA issue slots and SGPR-base loads (key 35 = 1 / 0), operator level + in-model.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06c; mkdir -p $O
for st in 1 0; do
  MID_AB_SHORT=1 MI355_TUNE="35=$st" timeout 600 python scripts/gemm_mid_ab.py > $O/gemm_mid_ab_stagger$st.txt 2>&1; echo "rc=$?" >> $O/gemm_mid_ab_stagger$st.txt
done
COMMON="--no-cpu-baseline --no-selfcheck --no-small-batch --no-clock-probe --no-families --no-train-step --no-vae"
for t in "32=0" "32=1" "32=1,35=0" "32=0" "32=1"; do
  MI355_TUNE="$t" timeout 300 python bench.py --steps 20 --warmup 3 --size 512 --batch 2 --guidance 4.5 --denoise-steps 10 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b2_512_cfg tune=$t', d['value'], d['ms_per_step'])" >> $O/inmodel_ab.txt
done
for t in "32=0" "32=1"; do
  MI355_TUNE="$t" timeout 300 python bench.py --steps 5 --warmup 2 --size 1024 --batch 1 --denoise-steps 28 $COMMON 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b1_1024 tune=$t', d['value'], d['ms_per_step'])" >> $O/inmodel_ab.txt
done
python - <<'PY'
import json,sys
for st in (1,0):
    print("stagger", st)
    for l in open(f"gpurun_out/r06c/gemm_mid_ab_stagger{st}.txt"):
        if l.startswith("{"):
            d=json.loads(l)
            if "M" in d: print(" ", d["M"],d["N"],d["K"], {k:(v["us_off"],v["us_forced"],v["bit_identical"]) for k,v in d.items() if isinstance(v,dict)})
            else: print(" ", d)
        elif "MID" in l or "rc=" in l or "rror" in l: print(" ", l.strip()[:300])
PY
cat $O/inmodel_ab.txt
