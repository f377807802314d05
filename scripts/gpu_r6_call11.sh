#!/bin/bash
# Round 6, call 11: after the stagger removal -- mid-kernel bit-identity script (short), GEMM / model / backward tests, then the evidence set.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O
MID_AB_SHORT=1 timeout 600 python scripts/gemm_mid_ab.py > $O/gemm_mid_ab.txt 2>&1; echo "rc=$?" >> $O/gemm_mid_ab.txt; tail -n 4 $O/gemm_mid_ab.txt | cut -c1-300
MI355_COMMIT=$MI355_COMMIT PARTS="bench stats1 pmc stats2 train small tests" bash scripts/gpu_r6_evidence.sh > $O/evidence.log 2>&1; tail -n 60 $O/evidence.log | cut -c1-250
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.txt 2>&1; tail -n 5 $O/smoke.txt | cut -c1-300
