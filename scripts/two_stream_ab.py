"""A/B of the two-stream SD3.5 forward (mi355_tune_set key 8) in ONE process: for every shape, the same rollout (hipGraph replay and eager
launches) with the text-stream chain on the caller's stream (mode 0) and on the plan's side stream (mode 1).  Results must be bit-identical;
wall-clock per rollout is compared.  Writes a recommendation for the auto mode (key 8 = 2, key 9 = rows): the largest image-stream row count
up to which two streams win on every measured shape.

    python scripts/two_stream_ab.py [--out gpurun_out/two_stream] [--quick]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import _lib  # noqa: E402
from mi355_flow.adapter import SD3_5NativeAdapter  # noqa: E402
from mi355_flow.engine import TransformerConfig  # noqa: E402
from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler  # noqa: E402
from mi355_flow.trajectory import compute_trajectory_indices  # noqa: E402
from mi355_flow.weights import synthetic_state_dict  # noqa: E402

N_TEXT = 333

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "two_stream"))
ap.add_argument("--quick", action="store_true", help="fewer timed rollouts")
ap.add_argument("--eager", action="store_true", help="also time eager launches (the shipped rollout replays the hipGraph)")
args = ap.parse_args()
os.makedirs(args.out, exist_ok=True)

lib = _lib.load()
dev = torch.device("cuda", 0)
cfg = TransformerConfig()
sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, dynamics_type="Flow-SDE", shift=3.0)
adapter = SD3_5NativeAdapter(synthetic_state_dict(cfg, device=dev, seed=1234), cfg, sched, latent_storage_dtype="fp16", device=dev)
adapter.rollout()

# (batch, size, guidance, denoise steps, timed rollouts): the reference's example shapes (512^2, N = 10) and the bench shape at B = 1..8
SHAPES = [(2, 512, 4.5, 10, 6), (8, 512, 1.0, 10, 6), (8, 512, 4.5, 10, 4), (1, 1024, 1.0, 28, 3), (2, 1024, 1.0, 28, 3), (4, 1024, 1.0, 28, 2),
          (8, 1024, 1.0, 28, 2)]
if os.environ.get("AB_SHAPES"):          # e.g. AB_SHAPES="2x512x4.5x10x6,8x1024x1x28x2"
    SHAPES = [tuple(float(v) if i == 2 else int(v) for i, v in enumerate(t.split("x"))) for t in os.environ["AB_SHAPES"].split(",")]
# mode 0 = one stream; 1 = text chain on a side stream, forked right after the joint attention; 2 = forked after the block's last attention
# (key 10 = 1); 3 = mode 1 + a third stream for the image V^T / dual-attention projections (key 11 = 1)
MODES = tuple(int(m) for m in os.environ.get("AB_MODES", "0,1,2").split(","))
NAMES = {0: "single", 1: "early", 2: "late", 3: "three"}
rows_out = []
log = open(os.path.join(args.out, "two_stream_ab.txt"), "w")


def say(s):
    print(s, flush=True)
    log.write(s + "\n")
    log.flush()


say("# two-stream forward A/B (scripts/two_stream_ab.py): ms per rollout (hipGraph replay); single = one stream, early = text chain on a side stream forked "
    "after the joint attention, late = forked after the block's last attention (shipped)")
say("# B  size  cfg  N   image rows " + "".join(f"{NAMES[m]:>10s}" for m in MODES) + "  " + "".join(f"{NAMES[m] + '%':>8s}" for m in MODES if m) +
    "   denoise-steps/s (single -> best)")
for (B, size, gs, N, iters) in SHAPES:
    if args.quick:
        iters = max(1, iters // 2)
    g = torch.Generator(device=dev).manual_seed(7)
    cfg_on = gs > 1.0
    pe = torch.randn(B, N_TEXT, cfg.joint_attention_dim, device=dev, generator=g).bfloat16()
    pp = torch.randn(B, cfg.pooled_projection_dim, device=dev, generator=g).bfloat16()
    ne = torch.randn(B, N_TEXT, cfg.joint_attention_dim, device=dev, generator=g).bfloat16() if cfg_on else None
    npl = torch.randn(B, cfg.pooled_projection_dim, device=dev, generator=g).bfloat16() if cfg_on else None
    traj = compute_trajectory_indices(sched.train_timesteps, N)

    def one(seed=None):
        if seed is not None:
            torch.cuda.manual_seed(seed)
            sched.set_seed(42)
        return adapter.inference(prompt=None, height=size, width=size, num_inference_steps=N, guidance_scale=gs, prompt_embeds=pe,
                                 pooled_prompt_embeds=pp, negative_prompt_embeds=ne, negative_pooled_prompt_embeds=npl,
                                 compute_log_prob=True, trajectory_indices=traj)

    res, ref = {}, None
    for graph in ((1, 0) if args.eager else (1,)):
        for mode in MODES:
            lib.mi355_tune_set(2, graph)
            lib.mi355_tune_set(8, 1 if mode else 0)
            lib.mi355_tune_set(10, 1 if mode == 2 else 0)
            lib.mi355_tune_set(11, 1 if mode == 3 else 0)
            s = one(seed=99)                      # eager warm-up (first call of a plan) or (re)capture
            s = one(seed=99)
            torch.cuda.synchronize()
            if ref is None:
                ref = s
            else:
                for a, b in zip(s, ref):
                    assert torch.equal(a.all_latents, b.all_latents), ("two-stream / graph changed the trajectory", B, size, graph, mode)
                    assert torch.equal(a.log_probs.nan_to_num(), b.log_probs.nan_to_num()), ("log-probs differ", B, size, graph, mode)
            t0 = time.perf_counter()
            for _ in range(iters):
                one()
            torch.cuda.synchronize()
            res[(graph, mode)] = (time.perf_counter() - t0) / iters * 1e3
    n_cfg = 2 if cfg_on else 1
    Mi = B * n_cfg * (size // 16) ** 2
    g0 = res[(1, 0)]
    cols = "".join(f" {res[(1, m)]:9.2f}" for m in MODES) + "  " + "".join(f" {(g0 / res[(1, m)] - 1) * 100:+7.2f}" for m in MODES if m)
    best = min(MODES, key=lambda m: res[(1, m)])
    say(f"{B:3d} {size:5d} {gs:4.1f} {N:3d} {Mi:10d}{cols}   {B * N / g0 * 1e3:8.2f} -> {B * N / res[(1, best)] * 1e3:8.2f} ({NAMES[best]})")
    rows_out.append(dict(batch=B, size=size, guidance=gs, denoise_steps=N, image_rows=Mi, ms={NAMES[m]: res[(1, m)] for m in MODES},
                         graph_gain_pct=(g0 / min(res[(1, m)] for m in MODES if m) - 1) * 100,
                         eager_ms={NAMES[k[1]]: v for k, v in res.items() if k[0] == 0}))

# recommendation: the largest row count R such that every measured shape with image_rows <= R gains >= 0.5 % in graph mode
rows_out.sort(key=lambda r: r["image_rows"])
rec = 0
for r in rows_out:
    if r["graph_gain_pct"] >= 0.5 and all(q["graph_gain_pct"] >= 0.5 for q in rows_out if q["image_rows"] <= r["image_rows"]):
        rec = r["image_rows"]
tune = f"8=2,9={rec},10=1" if rec > 0 else ""
say(f"# recommendation: auto mode up to {rec} image rows  ->  MI355_TUNE=\"{tune}\"")
with open(os.path.join(args.out, "two_stream_ab.json"), "w") as f:
    json.dump(dict(shapes=rows_out, recommended_rows=rec, tune=tune), f, indent=1)
with open(os.path.join(args.out, "tune.env"), "w") as f:
    f.write(tune)
lib.mi355_tune_set(8, 0)
lib.mi355_tune_set(10, 2)
lib.mi355_tune_set(11, 0)
lib.mi355_tune_set(2, 1)
log.close()
