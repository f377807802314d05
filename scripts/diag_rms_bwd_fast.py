"""Cross-check of the default-scope RMSNorm-backward gather (rms_bwd_gather_fast_kernel) against the general kernel at the real width (24 heads):
the same attention-projection gradients computed in the default scope (fast kernel) and in the full scope (general kernel, which also forms
the norm-weight partials) must agree to summation-order accuracy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flow-factory_amd")):
    sys.path.insert(0, p)
import torch
from mi355_flow.adapter import SD3_5NativeAdapter
from mi355_flow.engine import TransformerConfig
from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
from mi355_flow.weights import module_from_state_dict, synthetic_state_dict

dev = torch.device("cuda")
cfg = TransformerConfig()
mod = module_from_state_dict(synthetic_state_dict(cfg, device=dev, seed=1234))
ATTN = (".to_q.", ".to_k.", ".to_v.", ".to_out.0.")
sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, shift=3.0)
ad = SD3_5NativeAdapter(mod, cfg, sched, latent_storage_dtype="fp16")
ad.rollout()
B, lat, Nt = 1, 64, 333
g = torch.Generator(device=dev).manual_seed(1)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
sched.set_timesteps(28)
ts = sched.timesteps
kw = dict(t=ts[2].expand(B), t_next=ts[3].expand(B), latents=mk(B, 16, lat, lat).half(), next_latents=mk(B, 16, lat, lat).half(),
          prompt_embeds=mk(B, Nt, 4096).bfloat16(), pooled_prompt_embeds=mk(B, 2048).bfloat16(), guidance_scale=1.0, noise_level=0.7,
          compute_log_prob=True, return_kwargs=["log_prob", "dt"])


def grads(all_params):
    for n, p in mod.named_parameters():
        p.requires_grad_(all_params or (n.startswith("transformer_blocks.") and any(k in n for k in ATTN)))
        p.grad = None
    out = ad.forward(**kw)
    out.log_prob.sum().backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in mod.named_parameters() if n.startswith("transformer_blocks.") and any(k in n for k in ATTN)}


g1, g2 = grads(False), grads(True)
worst = max(float((g1[n] - g2[n]).norm() / (g2[n].norm() + 1e-30)) for n in g1)
print(f"{len(g1)} attention-projection gradients, default scope (fast gather) vs full scope (general gather): worst rel-L2 {worst:.3e}")
assert worst < 2e-3, worst
