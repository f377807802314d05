import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow.adapter import SD3_5NativeAdapter
from mi355_flow.engine import TransformerConfig
from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
from oracle import mmditx_ref as M
cfg_o = M.tiny_config(num_layers=3, num_heads=2, dual_layers=(0, 1), joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24)
sd = {k: v.bfloat16().float() for k, v in M.make_synthetic_state_dict(cfg_o, seed=1234, std=0.08).items()}
cfg_e = TransformerConfig(num_layers=3, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24, dual_layers=(0, 1))
sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=2, seed=42, dynamics_type="Flow-SDE", shift=3.0)
ad = SD3_5NativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg_e, sched, latent_storage_dtype="fp16")
ad.rollout()
B, Nt, N = 2, 13, 6
g = torch.Generator().manual_seed(6)
mk = lambda *s: torch.randn(*s, generator=g).bfloat16().cuda()
pe, pp, ne, npl = mk(B, Nt, 128), mk(B, 128), mk(B, Nt, 128), mk(B, 128)
torch.cuda.manual_seed(77)
s = ad.inference(prompt=None, height=128, width=128, num_inference_steps=N, guidance_scale=4.5, prompt_embeds=pe, pooled_prompt_embeds=pp,
                 negative_prompt_embeds=ne, negative_pooled_prompt_embeds=npl, trajectory_indices="all")
ts = s[0].timesteps
print("ts", ts.tolist(), "lp map", s[0].log_prob_index_map.tolist(), "lps", [x.log_probs.tolist() for x in s])
print("host eta", ad.scheduler.host_noise_levels(), "sigmas", ad.scheduler.sigmas.tolist())
plan = ad.engine.plan(B, 2, 16, 16, Nt, 1)
for i in (1, 2, 3):
    lat = torch.stack([x.all_latents[i] for x in s]); nxt = torch.stack([x.all_latents[i + 1] for x in s])
    t = ts[i].expand(B); tn = ts[i + 1].expand(B)
    out = ad.forward(t=t, t_next=tn, latents=lat, next_latents=nxt, prompt_embeds=pe, pooled_prompt_embeds=pp, negative_prompt_embeds=ne,
                     negative_pooled_prompt_embeds=npl, guidance_scale=4.5, noise_level=0.7, return_kwargs=["log_prob", "std_dev_t", "dt"])
    print(i, "adapter.forward", out.log_prob.tolist(), out.std_dev_t.flatten().tolist(), out.dt.flatten().tolist(), "sigma gpu", (t/1000).tolist())
    th = float(ts[i]); tnh = float(ts[i + 1])
    o = plan.denoise_step(lat, torch.tensor(th), ne, npl, pe, pp, 4.5, torch.tensor(th) / 1000, torch.tensor(tnh) / 1000, 0.7, float(sched.sigmas[1]),
                          "Flow-SDE", next_latents=nxt, want=("std_dev_t", "dt"))
    print(i, "direct scalar  ", o.log_prob.tolist(), o.std_dev_t.tolist(), o.dt.tolist())
    o = plan.denoise_step(lat, torch.full((B,), th), ne, npl, pe, pp, 4.5, torch.full((B,), th) / 1000, torch.full((B,), tnh) / 1000, torch.full((B,), 0.7),
                          float(sched.sigmas[1]), "Flow-SDE", next_latents=nxt, want=("std_dev_t", "dt"))
    print(i, "direct per-samp", o.log_prob.tolist(), o.std_dev_t.tolist(), o.dt.tolist())
