#!/bin/bash
# round 4, fourth GPU call: attention128 backward after the loop restructure (numerics + per-kernel time), FLUX step, RCCL test, text-GEMM dispatch A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_attn -o t -- python $GRAFT_REPO_ROOT/scripts/attn128_bwd_ab.py > $O/attn128_bwd_ab.txt 2>&1)
grep -E "attn128|Name" $O/prof_attn/*kernel_stats.csv | cut -c1-200 >> $O/attn128_bwd_ab.txt
timeout 600 python -m pytest tests/test_gpu_flux_backward.py -x -q -m gpu -k "not full_width" > $O/pytest_flux_backward.txt 2>&1; echo "rc=$?" >> $O/pytest_flux_backward.txt
timeout 300 python -m pytest tests/test_gpu_ddp_rccl.py -x -q -s -m gpu > $O/pytest_ddp_rccl.txt 2>&1; echo "rc=$?" >> $O/pytest_ddp_rccl.txt
timeout 600 python scripts/flux_train_bench.py --batch 1 --size 1024 > $O/flux_train_b1_1024.json 2>/dev/null
for T in "" "3=64" "3=32"; do
  MI355_TUNE="$T" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-train-step --no-families --no-clock-probe --no-selfcheck 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'tune': '$T', 'value': d['value'], 'forward_frac': d['roofline']['forward']['frac'], 'small': {k: v.get('denoise_steps_per_s') for k, v in d.get('small_batch', {}).items() if isinstance(v, dict)}}))" >> $O/text_gemm_dispatch_ab.txt
done
find $O -type f -size +1M -delete
tail -n 6 $O/*.txt $O/*.json | cut -c1-1200
