#!/bin/bash
# Round 6, call 14: first contact of the 256x192-tile GEMM kernel (gemm_w6_kernel, key 40): bit identity + op timing + whole-forward A/B,
# then small-batch bench line B = 2, 1024^2 under key 40 = 0 / 1, and the optimize() step A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06n; mkdir -p $O
( time timeout 900 python scripts/gemm_w6_ab.py ) > $O/gemm_w6_ab.txt 2>&1; echo "rc=$?" >> $O/gemm_w6_ab.txt
cut -c1-420 $O/gemm_w6_ab.txt | tail -n 30
for i in 1 2 3; do for t in "40=0" "40=1"; do
  MI355_TUNE="$t" timeout 300 python scripts/train_bench.py --batch 2 --size 1024 --train default --iters 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('sd3 train b2_1024 tune=$t', d['ms_forward_backward'], d['ms_forward_train'], d.get('ms_forward_nograd'))" >> $O/train_w6_ab.txt
done; done
cat $O/train_w6_ab.txt
