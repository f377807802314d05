"""How much of the rollout's speed is data-dependent power?  (measurement tool, not the product path)

The SD3.5 rollout runs at the package power cap (bench.py `power`: ~1365 W of 1400; delivered clock ~2.1 of 2.4 GHz).  DESIGN.md 14.2 found that
skipping kernels whose OUTPUT feeds the matrix pipes zeros made the whole rollout 10-18 % faster: MFMA power depends on operand toggling.  This
script measures that ceiling directly: the same rollout (same launches, same shapes, same schedules, hipGraph replay) on

    scale 1.0   the synthetic weights of bench.py (random bf16 operands everywhere)
    scale 0.0   all-zero weights: every activation behind the first GEMM is zero -- the kernels execute the same instruction streams with
                (almost) no operand toggling; instruction issue, LDS / HBM traffic and launch structure are unchanged
    scale s     anything between (smaller magnitudes toggle the same mantissa bits: expected ~ scale 1.0)

and reports denoise-steps/s, socket power and the PLL clock for each.  What scale 0.0 reaches is what the instruction schedules of today's
kernels deliver when the power cap does not bind; the gap to scale 1.0 is what only fewer joules per FLOP can recover.

    python scripts/power_ceiling.py [--batch 8] [--size 1024] [--denoise-steps 28] [--rollouts 3] [--scales 1.0,0.0]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flow-factory_amd"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--denoise-steps", type=int, default=28)
ap.add_argument("--rollouts", type=int, default=3)
ap.add_argument("--scales", default="1.0,0.0")
args = ap.parse_args()

from mi355_flow.adapter import SD3_5NativeAdapter  # noqa: E402
from mi355_flow.engine import TransformerConfig  # noqa: E402
from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler  # noqa: E402
from mi355_flow.trajectory import compute_trajectory_indices  # noqa: E402
from mi355_flow.weights import synthetic_state_dict  # noqa: E402
from power_meter import PowerMeter  # noqa: E402

dev = torch.device("cuda")
cfg = TransformerConfig()
B, N = args.batch, args.denoise_steps
g = torch.Generator(device=dev).manual_seed(0)
pe = torch.randn(B, 333, cfg.joint_attention_dim, device=dev, generator=g).bfloat16()
pp = torch.randn(B, cfg.pooled_projection_dim, device=dev, generator=g).bfloat16()
for scale in [float(s) for s in args.scales.split(",")]:
    sd = synthetic_state_dict(cfg, device=dev, seed=1234)
    if scale != 1.0:
        sd = {k: (v * scale if v.is_floating_point() else v) for k, v in sd.items()}
    sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, dynamics_type="Flow-SDE", shift=3.0)
    ad = SD3_5NativeAdapter(sd, cfg, sched, latent_storage_dtype="fp16", device=dev)
    ad.rollout()
    traj = compute_trajectory_indices(sched.train_timesteps, N)

    def one():
        return ad.inference(prompt=None, height=args.size, width=args.size, num_inference_steps=N, guidance_scale=1.0, prompt_embeds=pe * scale,
                            pooled_prompt_embeds=pp * scale, compute_log_prob=True, trajectory_indices=traj)
    one(); one()                                     # eager warm-up + graph capture
    torch.cuda.synchronize()
    out = {"weight_scale": scale, "batch": B, "size": args.size, "denoise_steps": N}
    try:
        with PowerMeter(device=0, period_s=0.02) as pm:
            t0 = time.perf_counter()
            for _ in range(args.rollouts):
                s = one()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        ps = pm.summary()
        out["power_w_median"] = (ps.get("watts") or {}).get("median")
        out["sclk_mhz_median"] = (ps.get("sclk_mhz_reported") or {}).get("median")
        out["joules_per_denoise_step"] = round(ps["energy_j"] / (args.rollouts * B * N), 2) if ps.get("energy_j") else None
    except Exception as e:  # noqa: BLE001
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.rollouts):
            s = one()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["power_error"] = repr(e)
    out["denoise_steps_per_s"] = round(args.rollouts * B * N / dt, 2)
    out["finite"] = bool(torch.isfinite(s[0].all_latents.float()).all())
    print(json.dumps(out), flush=True)
    ad.engine.close()
    del ad, sd
    torch.cuda.empty_cache()
