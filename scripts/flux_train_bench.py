"""Times the native FLUX.1 `optimize()` replay step (SURVEY.md 8(f) N1 over N3; reference trainers/grpo.py:263, :326-330 over
models/flux/flux1.py:294-346) at the FLUX.1-dev geometry (11.9 B parameters, synthetic bf16 master weights): no-grad replay forward,
grad-mode forward (activation stash) and forward + backward, for the reference's default target modules (flux1.py:76-84).

    python scripts/flux_train_bench.py [--batch 1] [--size 1024] [--n-text 512] [--iters 2] [--layers 19 --single-layers 38]

Algorithmic FLOPs (2 FLOP/MAC, matmuls only): forward F; backward = data gradients of every block linear (= their forward FLOPs) + attention
backward (2.5 x attention forward: 5 tile products vs 2) + weight gradients of the TRAINABLE linears (= their forward FLOPs).  The
implementation spends 3.5 x in the attention backward (two deterministic passes, 7 tile products)."""
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
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--n-text", type=int, default=512)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--layers", type=int, default=19)
ap.add_argument("--single-layers", type=int, default=38)
ap.add_argument("--only-step", action="store_true", help="run 1 + iters forward+backward steps and nothing else (for rocprofv3)")
args = ap.parse_args()

from mi355_flow import flux  # noqa: E402
from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler  # noqa: E402
from mi355_flow.weights import module_from_state_dict, synthetic_flux_state_dict  # noqa: E402

dev = torch.device("cuda")
cfg = flux.FluxConfig(num_layers=args.layers, num_single_layers=args.single_layers)
mod = module_from_state_dict(synthetic_flux_state_dict(cfg, device=dev), buffers=())
DEFAULT = ("attn.to_k.", "attn.to_q.", "attn.to_v.", "attn.to_out.0.", "attn.add_k_proj.", "attn.add_q_proj.", "attn.add_v_proj.", "attn.to_add_out.",
           "ff.net.0.proj.", "ff.net.2.", "ff_context.net.0.proj.", "ff_context.net.2.")
n_train = 0
for n, p in mod.named_parameters():
    on = any(k in n for k in DEFAULT)
    p.requires_grad_(on)
    n_train += p.numel() if on else 0
sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, shift=3.0)
ad = flux.Flux1NativeAdapter(mod, cfg, sched, latent_storage_dtype="fp16")
ad.rollout()
B, Nt = args.batch, args.n_text
h = w = args.size // 8
Ni = (h // 2) * (w // 2)
g = torch.Generator(device=dev).manual_seed(1)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
sched.set_timesteps(28)
ts = sched.timesteps
kw = dict(t=ts[2].expand(B), t_next=ts[3].expand(B), latents=mk(B, Ni, 64).half(), next_latents=mk(B, Ni, 64).half(),
          prompt_embeds=mk(B, Nt, cfg.joint_attention_dim).bfloat16(), pooled_prompt_embeds=mk(B, cfg.pooled_projection_dim).bfloat16(),
          height=args.size, width=args.size, guidance_scale=3.5, noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "dt"])


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
ratio_is_one = bool(torch.equal(lp_a, lp_b))
D, S, L, LS = cfg.dim, Ni + Nt, cfg.num_layers, cfg.num_single_layers
lin = (L + LS) * S * 12 * D * D                     # block linears: q k v o (4 D^2) + MLP (8 D^2) per token and block
attn = (L + LS) * 2 * S * S * D
lin_train = L * S * 12 * D * D + LS * S * 3 * D * D  # default targets: everything in the double blocks, q k v in the single blocks
fwd_fl = 2.0 * (lin + attn) * B
bwd_fl = 2.0 * (lin + 2.5 * attn + lin_train) * B
plan = ad.engine.plan(B, h, w, Nt, 1)
print(json.dumps({
    "what": "GRPO optimize() replay step, FLUX.1 geometry, synthetic weights", "layers": [L, LS], "batch": B, "size": args.size, "tokens": S,
    "trainable": "default target modules (flux1.py:76-84)", "trainable_params": n_train,
    "ms_forward_nograd": round(t_ng * 1e3, 2), "ms_forward_train": round(t_f * 1e3, 2), "ms_forward_backward": round(t_fb * 1e3, 2),
    "ms_backward": round((t_fb - t_f) * 1e3, 2),
    "tflops_forward_train": round(fwd_fl / t_f / 1e12, 1), "tflops_backward": round(bwd_fl / (t_fb - t_f) / 1e12, 1),
    "tflops_step": round((fwd_fl + bwd_fl) / t_fb / 1e12, 1), "frac_of_2500": round((fwd_fl + bwd_fl) / t_fb / 2.5e15, 4),
    "stash_plus_scratch_GiB": round(plan.training_bytes / 2 ** 30, 2), "hbm_in_use_GiB": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2 ** 30, 1),
    "ratio_is_one": ratio_is_one}))
