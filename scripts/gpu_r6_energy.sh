#!/bin/bash
# Round 6: ONE energy experiment at the headline shape, judged by power.joules_per_denoise_step.  Lever:
# 128 x 128 REGISTER tiles for the N = 1536 GEMMs -- the 4-wave hand-scheduled kernel (a wave's fragment feeds 8 MFMAs: fewer LDS bytes per
# MFMA) also for the gated-residual / V^T shapes the default dispatch leaves on the 8-wave ping-pong kernel (key 0 = 2).  Same lease, interleaved.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O
LEAN="--no-cpu-baseline --no-small-batch --no-vae --no-families --no-train-step --no-selfcheck"
for t in "0=1" "0=2" "0=1" "0=2"; do
  MI355_TUNE="$t" timeout 600 python bench.py --steps 5 --warmup 2 $LEAN 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('power',{}); r=d['roofline']
print(json.dumps({'tune':'$t','value':d['value'],'ms_per_step':d['ms_per_step'],'forward_frac':r['forward']['frac'],'gemm_frac':r.get('gemm',{}).get('frac'),
 'joules_per_denoise_step':p.get('joules_per_denoise_step'),'watts_median':p.get('watts',{}).get('median'),'cap_watts':p.get('cap_watts'),
 'sclk_mhz_reported_median':p.get('sclk_mhz_reported',{}).get('median'),'delivered_clock_mhz':r.get('delivered_clock_mhz'),'algorithmic_tflop_per_joule':p.get('algorithmic_tflop_per_joule')}))" >> $O/energy_ab.jsonl
done
cat $O/energy_ab.jsonl | cut -c1-700
