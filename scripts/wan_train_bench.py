"""Times the native Wan `optimize()` replay step (SURVEY.md 8(f) N1 over N4; reference trainers/grpo.py:263, :326-330 over
models/wan/wan2_t2v.py:426-543) at the Wan2.1-T2V-1.3B geometry (30 layers, synthetic bf16 master weights, CFG = forward batch [negative | positive]):
no-grad replay forward, grad-mode forward (activation stash) and forward + backward, for the reference's default target modules (wan2_t2v.py:74-85).
NOT YET RUN ON A GPU (written after round 4's GPU budget was spent): round 5's first call.

    python scripts/wan_train_bench.py [--batch 1] [--frames 49] [--height 480] [--width 832] [--n-text 512] [--guidance 5.0] [--iters 2] [--layers 30]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flow-factory_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--frames", type=int, default=49)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=832)
ap.add_argument("--n-text", type=int, default=512)
ap.add_argument("--guidance", type=float, default=5.0)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--layers", type=int, default=30)
ap.add_argument("--only-step", action="store_true", help="run 1 + iters forward+backward steps and nothing else (for rocprofv3)")
args = ap.parse_args()

from mi355_flow import wan  # noqa: E402
from mi355_flow.weights import module_from_state_dict, synthetic_wan_state_dict  # noqa: E402

dev = torch.device("cuda")
cfg = wan.WanConfig(num_layers=args.layers)
mod = module_from_state_dict(synthetic_wan_state_dict(cfg, device=dev), buffers=())
DEFAULT = (".attn1.to_q.", ".attn1.to_k.", ".attn1.to_v.", ".attn1.to_out.0.", ".attn2.to_q.", ".attn2.to_k.", ".attn2.to_v.", ".attn2.to_out.0.",
           ".ffn.net.0.proj.", ".ffn.net.2.")
n_train = 0
for n, p in mod.named_parameters():
    on = any(k in n for k in DEFAULT)
    p.requires_grad_(on)
    n_train += p.numel() if on else 0
sched = wan.UniPCMultistepSDEScheduler(flow_shift=3.0, noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42)
ad = wan.Wan2T2VNativeAdapter(mod, cfg, sched, latent_storage_dtype="fp16")
ad.rollout()
B, Nt = args.batch, args.n_text
T, h, w = (args.frames - 1) // 4 + 1, args.height // 8, args.width // 8
g = torch.Generator(device=dev).manual_seed(1)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
sched.set_timesteps(28)
ts = sched.timesteps
kw = dict(t=ts[2].float().expand(B), t_next=ts[3].float().expand(B), latents=mk(B, 16, T, h, w).half(), next_latents=mk(B, 16, T, h, w).half(),
          prompt_embeds=mk(B, Nt, cfg.text_dim).bfloat16(), guidance_scale=args.guidance, noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "dt"])
n_cfg = 2 if args.guidance > 1 else 1
if n_cfg == 2:
    kw["negative_prompt_embeds"] = mk(B, Nt, cfg.text_dim).bfloat16()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def nograd():
    with torch.no_grad():
        return ad.forward(**kw)


def fwd():
    return ad.forward(**kw)


def fwd_bwd():
    out = ad.forward(**kw)
    out.log_prob.sum().backward()
    for p in mod.parameters():
        p.grad = None


if args.only_step:
    print(json.dumps({"ms_forward_backward": round(timed(fwd_bwd, args.iters) * 1e3, 2), "steps_profiled": args.iters + 1}))
    sys.exit(0)
t_ng, t_f, t_fb = timed(nograd, args.iters), timed(fwd, args.iters), timed(fwd_bwd, args.iters)
lp_a, lp_b = nograd().log_prob, ad.forward(**kw).log_prob.detach()
D, F, L, Bp = cfg.dim, cfg.ffn_dim, cfg.num_layers, B * n_cfg
S = T * (h // 2) * (w // 2)
lin = L * S * (6 * D * D + 2 * D * F)                       # per token and block: attn1 q k v o + attn2 q o (6 D^2; text k / v are per text token) + FFN
lin += L * Nt * 2 * D * D
attn = L * (2 * S * S * D + 2 * S * Nt * D)
fwd_fl = 2.0 * (lin + attn) * Bp
bwd_fl = 2.0 * (2 * lin + 2.5 * attn) * Bp                  # data gradients + weight gradients of every block linear (all trainable) + attention backward
plan = next(iter(ad.engine._plans.values()))
print(json.dumps({
    "what": "GRPO optimize() replay step, Wan2.1-T2V-1.3B geometry, synthetic weights", "layers": L, "batch": B, "n_cfg": n_cfg,
    "clip": [args.frames, args.height, args.width], "tokens": S, "trainable": "default target modules (wan2_t2v.py:74-85)", "trainable_params": n_train,
    "ms_forward_nograd": round(t_ng * 1e3, 2), "ms_forward_train": round(t_f * 1e3, 2), "ms_forward_backward": round(t_fb * 1e3, 2),
    "ms_backward": round((t_fb - t_f) * 1e3, 2), "tflops_forward_train": round(fwd_fl / t_f / 1e12, 1),
    "tflops_backward": round(bwd_fl / (t_fb - t_f) / 1e12, 1), "tflops_step": round((fwd_fl + bwd_fl) / t_fb / 1e12, 1),
    "frac_of_2500": round((fwd_fl + bwd_fl) / t_fb / 2.5e15, 4), "stash_plus_scratch_GiB": round(plan.training_bytes / 2 ** 30, 2),
    "hbm_in_use_GiB": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2 ** 30, 1), "ratio_is_one": bool(torch.equal(lp_a, lp_b))}))
