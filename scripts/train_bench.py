"""Times the differentiable replay step of GRPO's optimize() (reference trainers/grpo.py:229-330) on the full
SD3.5-medium geometry: no-grad replay forward, grad-mode forward (activation stash) and backward, for a trainable set.

    python scripts/train_bench.py [--batch 2] [--size 1024] [--train default|attn|blocks] [--guidance 1.0] [--iters 3]

--train default = SD3_5Adapter.default_target_modules (reference models/stable_diffusion/sd3_5.py:75-80): the eight "attn.*" projections,
image AND text side, matched by substring (models/abc.py:1793) -- "attn2.*" does not match.  --train attn = the base class's set
(models/abc.py:382-385: to_q / to_k / to_v / to_out.0, which also matches attn2).

Algorithmic FLOPs (2 FLOP/MAC, matmuls only): forward F; backward = data gradients of every linear (= their forward
FLOPs) + attention backward (2.5 x attention forward: 5 tile products vs 2) + weight gradients of the TRAINABLE linears (= their
forward FLOPs).  The implementation spends 3.5 x in the attention backward (two deterministic passes, 7 tile products)."""
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
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--train", choices=["default", "attn", "blocks"], default="default")
ap.add_argument("--guidance", type=float, default=1.0)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--only-step", action="store_true", help="run 1 + iters forward+backward steps and nothing else (for rocprofv3)")
args = ap.parse_args()

from mi355_flow.adapter import SD3_5NativeAdapter  # noqa: E402
from mi355_flow.engine import TransformerConfig  # noqa: E402
from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler  # noqa: E402
from mi355_flow.weights import module_from_state_dict, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda")
cfg = TransformerConfig()
mod = module_from_state_dict(synthetic_state_dict(cfg, device=dev, seed=1234))
DEFAULT = ("attn.add_q_proj", "attn.add_k_proj", "attn.add_v_proj", "attn.to_add_out",     # SD3_5Adapter.default_target_modules
           "attn.to_q", "attn.to_k", "attn.to_v", "attn.to_out.0")                          # (sd3_5.py:75-80), substring match
ATTN = (".to_q.", ".to_k.", ".to_v.", ".to_out.0.")           # BaseAdapter.default_target_modules (models/abc.py:382-385), incl. attn2
BLOCKS = ATTN + (".add_q_proj.", ".add_k_proj.", ".add_v_proj.", ".to_add_out.", ".ff.net.", ".ff_context.net.")
keys = {"default": DEFAULT, "attn": ATTN, "blocks": BLOCKS}[args.train]
n_train = 0
for n, p in mod.named_parameters():
    on = any(k in n for k in keys) if args.train == "default" else (n.startswith("transformer_blocks.") and any(k in n for k in keys))
    p.requires_grad_(on)
    n_train += p.numel() if on else 0
sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, shift=3.0)
ad = SD3_5NativeAdapter(mod, cfg, sched, latent_storage_dtype="fp16")
ad.rollout()
B, lat, Nt = args.batch, args.size // 8, 333
g = torch.Generator(device=dev).manual_seed(1)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
cfg_on = args.guidance > 1.0
sched.set_timesteps(28)
ts = sched.timesteps
kw = dict(t=ts[2].expand(B), t_next=ts[3].expand(B), latents=mk(B, 16, lat, lat).half(), next_latents=mk(B, 16, lat, lat).half(),
          prompt_embeds=mk(B, Nt, 4096).bfloat16(), pooled_prompt_embeds=mk(B, 2048).bfloat16(),
          negative_prompt_embeds=mk(B, Nt, 4096).bfloat16() if cfg_on else None,
          negative_pooled_prompt_embeds=mk(B, 2048).bfloat16() if cfg_on else None, guidance_scale=args.guidance, noise_level=0.7,
          compute_log_prob=True, return_kwargs=["log_prob", "dt"])


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


state = {}


def fwd():
    state["out"] = ad.forward(**kw)


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
assert torch.equal(lp_a, lp_b), "grad-mode replay log-prob differs from the no-grad replay"
D, F, L, Ld, Ni = cfg.dim, cfg.ff_mult * cfg.dim, cfg.num_layers, len(cfg.dual_layers), (lat // 2) ** 2
n_cfg = 2 if cfg_on else 1
lin_img = L * Ni * (4 * D * D + 2 * D * F) + Ld * Ni * 4 * D * D
lin_ctx = (L - 1) * Nt * (4 * D * D + 2 * D * F) + Nt * 3 * D * D
attn = L * 2 * (Ni + Nt) ** 2 * D + Ld * 2 * Ni * Ni * D
lin_train = {"attn": (L + Ld) * Ni * 4 * D * D,
             "default": L * Ni * 4 * D * D + (L - 1) * Nt * 4 * D * D + Nt * 3 * D * D,   # the last block has no to_add_out
             "blocks": lin_img + lin_ctx}[args.train]
fwd_fl = 2.0 * (lin_img + lin_ctx + attn) * B * n_cfg
bwd_fl = 2.0 * (lin_img + lin_ctx + 2.5 * attn + lin_train) * B * n_cfg
plan = ad.engine.plan(B, n_cfg, lat, lat, Nt, 1)
print(json.dumps({
    "what": "GRPO optimize() replay step, SD3.5-medium, synthetic weights", "batch": B, "size": args.size, "n_cfg": n_cfg,
    "trainable": args.train, "trainable_params": n_train,
    "ms_forward_nograd": round(t_ng * 1e3, 2), "ms_forward_train": round(t_f * 1e3, 2), "ms_forward_backward": round(t_fb * 1e3, 2),
    "ms_backward": round((t_fb - t_f) * 1e3, 2),
    "tflops_forward_train": round(fwd_fl / t_f / 1e12, 1), "tflops_backward": round(bwd_fl / (t_fb - t_f) / 1e12, 1),
    "tflops_step": round((fwd_fl + bwd_fl) / t_fb / 1e12, 1), "frac_of_2500": round((fwd_fl + bwd_fl) / t_fb / 2.5e15, 4),
    "stash_plus_scratch_GiB": round(plan.training_bytes / 2 ** 30, 2), "ratio_is_one": True}))
