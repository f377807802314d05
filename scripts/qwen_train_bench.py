"""Times the native Qwen-Image `optimize()` replay step (SURVEY.md 8(f) N1 over N4; reference trainers/grpo.py:263, :326-330 and
trainers/dgpo.py:352-364 over models/qwen_image/qwen_image.py:476-600) at the Qwen-Image geometry (60 layers, 20.4 B parameters, synthetic bf16
master weights, true CFG = forward batch [negative | positive]): no-grad replay forward, grad-mode forward (activation stash) and
forward + backward, for the reference's default target modules (qwen_image.py:81-89).

    python scripts/qwen_train_bench.py [--batch 1] [--size 1024] [--n-text 64] [--guidance 4.0] [--iters 2] [--layers 60]

Algorithmic FLOPs (2 FLOP/MAC, matmuls only) as scripts/flux_train_bench.py: forward F; backward = data gradients of every block linear +
attention backward (2.5 x attention forward) + weight gradients of the TRAINABLE linears."""
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
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--n-text", type=int, default=64)
ap.add_argument("--guidance", type=float, default=4.0)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--layers", type=int, default=60)
ap.add_argument("--only-step", action="store_true", help="run 1 + iters forward+backward steps and nothing else (for rocprofv3)")
args = ap.parse_args()

from mi355_flow import qwen  # noqa: E402
from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler  # noqa: E402
from mi355_flow.weights import module_from_state_dict  # noqa: E402
from qwen_bench import synthetic_tensor  # noqa: E402

dev = torch.device("cuda")
cfg = qwen.QwenConfig(num_layers=args.layers)
probe = qwen.QwenEngine(qwen.QwenConfig(num_layers=1))
names1 = probe.param_names()
probe.close()
names = [n for n in names1 if not n.startswith("transformer_blocks.")]
for i in range(args.layers):
    names += [n.replace("transformer_blocks.0.", f"transformer_blocks.{i}.") for n in names1 if n.startswith("transformer_blocks.0.")]
g = torch.Generator(device=dev).manual_seed(7)
mod = module_from_state_dict({n: synthetic_tensor(cfg, n, dev, g) for n in names}, buffers=())
DEFAULT = (".attn.to_q.", ".attn.to_k.", ".attn.to_v.", ".attn.to_out.0.", ".attn.add_q_proj.", ".attn.add_k_proj.", ".attn.add_v_proj.",
           ".attn.to_add_out.", ".img_mlp.net.0.proj.")
n_train = 0
for n, p in mod.named_parameters():
    on = any(k in n for k in DEFAULT)
    p.requires_grad_(on)
    n_train += p.numel() if on else 0
sched = FlowMatchEulerDiscreteSDEScheduler(shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, max_image_seq_len=8192,
                                           shift_terminal=0.02, sde_steps=[1, 2, 3], num_sde_steps=1, noise_level=0.7, seed=42)
ad = qwen.QwenImageNativeAdapter(mod, cfg, sched, latent_storage_dtype="bf16")
ad.rollout()
B, Nt = args.batch, args.n_text
hp = wp = args.size // 16
Ni = hp * wp
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
sched.set_timesteps(28, mu=0.8)
ts = sched.timesteps
n_cfg = 2 if args.guidance > 1 else 1
n_neg = max(1, Nt // 8)
kw = dict(t=ts[2].expand(B), t_next=ts[3].expand(B), latents=mk(B, Ni, 64).bfloat16(), next_latents=mk(B, Ni, 64).bfloat16(),
          prompt_embeds=mk(B, Nt, cfg.joint_attention_dim).bfloat16(), prompt_embeds_mask=torch.ones(B, Nt, dtype=torch.long, device=dev),
          img_shapes=[[(1, hp, wp)]] * B, guidance_scale=args.guidance, noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "dt"])
if n_cfg == 2:
    kw.update(negative_prompt_embeds=mk(B, n_neg, cfg.joint_attention_dim).bfloat16(),
              negative_prompt_embeds_mask=torch.ones(B, n_neg, dtype=torch.long, device=dev))


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
plan = next(iter(ad.engine._plans.values()))
D, L, FB = cfg.dim, cfg.num_layers, B * n_cfg
S = Ni + plan.n_text
lin = L * S * 12 * D * D - (Nt and plan.n_text * 8 * D * D)          # block linears per token: q k v o (4 D^2) + MLP (8 D^2); last block: no text MLP
attn = L * 2 * S * S * D
lin_train = L * (S * 4 * D * D + Ni * 4 * D * D)                     # default targets: attention projections of both streams + img_mlp.net.0.proj
fwd_fl = 2.0 * (lin + attn) * FB
bwd_fl = 2.0 * (lin + 2.5 * attn + lin_train) * FB
print(json.dumps({
    "what": "optimize() replay step, Qwen-Image geometry, synthetic weights", "layers": L, "batch": B, "n_cfg": n_cfg, "size": args.size, "tokens": S,
    "trainable": "default target modules (qwen_image.py:81-89)", "trainable_params": n_train,
    "ms_forward_nograd": round(t_ng * 1e3, 2), "ms_forward_train": round(t_f * 1e3, 2), "ms_forward_backward": round(t_fb * 1e3, 2),
    "ms_backward": round((t_fb - t_f) * 1e3, 2),
    "tflops_forward_train": round(fwd_fl / t_f / 1e12, 1), "tflops_backward": round(bwd_fl / (t_fb - t_f) / 1e12, 1),
    "tflops_step": round((fwd_fl + bwd_fl) / t_fb / 1e12, 1), "frac_of_2500": round((fwd_fl + bwd_fl) / t_fb / 2.5e15, 4),
    "stash_plus_scratch_GiB": round(plan.training_bytes / 2 ** 30, 2),
    "hbm_in_use_GiB": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2 ** 30, 1), "ratio_is_one": ratio_is_one}))
