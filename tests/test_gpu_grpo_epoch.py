"""GPU integration test: one GRPO epoch with BOTH halves on the engine, written the way the reference trainer does it
(src/flow_factory/trainers/grpo.py: sample() :141-173, compute_advantages, optimize() :185-342) --

  sample      adapter.rollout(); adapter.inference(...)                        -> samples with (x_i, x_{i+1}), old log-probs
  advantages  group-normalised rewards (mi355_flow.advantage, the reference's GDPO / weighted-sum arithmetic)
  optimize    per trained timestep: adapter.forward(t, latents, next_latents, ...) WITH autograd -> ratio -> PPO-clip loss -> backward ->
              clip_grad_norm -> AdamW step

and checks the invariants that make this a drop-in: the first ratio of every sample is EXACTLY 1 (default clip_range +-1e-4 would otherwise
clip from the first step on), gradients are finite and non-zero, the update changes the policy (second pass ratio != 1), and the next
rollout runs on the updated weights without any explicit re-bind (weights are live)."""
import numpy as np
import pytest
import torch

import _plugin_fakes as PF

pytestmark = pytest.mark.gpu


def test_one_grpo_epoch_rollout_and_optimize_on_the_engine():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import advantage as ADV
    from mi355_flow.adapter import SD3_5NativeAdapter
    from mi355_flow.engine import TransformerConfig
    from mi355_flow.samples import BaseSample
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    from mi355_flow.trajectory import compute_trajectory_indices
    from mi355_flow.weights import expected_shapes
    cfg = TransformerConfig(num_layers=3, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24, dual_layers=(0, 1))
    mod = PF.build_module_tree(expected_shapes(cfg), seed=4, std=0.08).cuda().bfloat16()        # bf16 master weights (model_args.py:49)
    targets = (".to_q.", ".to_k.", ".to_v.", ".to_out.0.")                                        # default target modules
    for n, p in mod.named_parameters():
        p.requires_grad_(any(k in n for k in targets))
    sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=2, seed=42, shift=3.0)
    ad = SD3_5NativeAdapter(mod, cfg, sched, latent_storage_dtype="fp16")
    opt = torch.optim.AdamW([p for p in mod.parameters() if p.requires_grad], lr=2e-3)
    M, K, N, Nt, H = 2, 4, 6, 13, 128          # 2 prompts x 4 repeats, 6 denoise steps
    g = torch.Generator().manual_seed(0)
    pe_u, pp_u = torch.randn(M, Nt, 128, generator=g).bfloat16().cuda(), torch.randn(M, 128, generator=g).bfloat16().cuda()
    prompts = [f"prompt {i}" for i in range(M) for _ in range(K)]
    pe, pp = pe_u.repeat_interleave(K, 0), pp_u.repeat_interleave(K, 0)

    def sample():
        ad.rollout()
        traj = compute_trajectory_indices(ad.scheduler.train_timesteps, N)
        out = []
        for s in range(0, M * K, 4):            # micro-batches of 4
            out += ad.inference(prompt=prompts[s:s + 4], height=H, width=H, num_inference_steps=N, guidance_scale=1.0, prompt_embeds=pe[s:s + 4],
                                pooled_prompt_embeds=pp[s:s + 4], compute_log_prob=True, trajectory_indices=traj)
        return out

    torch.cuda.manual_seed(123)
    samples = sample()
    assert len(samples) == M * K and len({s.unique_id for s in samples}) == M
    # stand-in reward: negative mean of the final latent (any deterministic function of the sample)
    rewards = {"r": np.array([-float(s.all_latents[-1].float().mean()) for s in samples])}
    adv = ADV.compute_weighted_sum(rewards, {"r": 1.0}, [s.unique_id for s in samples], group_size=K, global_std=True)
    assert abs(float(adv.sum())) < 1e-6 and float(adv.abs().max()) > 0
    for s, a in zip(samples, adv):
        s.extra_kwargs["advantage"] = a.float().cuda()

    clip = 1e-4
    ad.train()
    first_ratios, second_ratios, grad_norms = [], [], []
    for inner in range(2):                      # second pass over the same samples: the policy has moved
        for start in range(0, M * K, 4):
            batch = BaseSample.stack(samples[start:start + 4])
            lmap, pmap = batch["latent_index_map"], batch["log_prob_index_map"]
            for t_idx in ad.scheduler.train_timesteps.tolist():
                old_lp = batch["log_probs"][:, pmap[t_idx]]
                t = batch["timesteps"][:, t_idx]
                t_next = batch["timesteps"][:, t_idx + 1] if t_idx + 1 < N else torch.zeros_like(t)
                out = ad.forward(t=t, t_next=t_next, latents=batch["all_latents"][:, lmap[t_idx]], next_latents=batch["all_latents"][:, lmap[t_idx + 1]],
                                 prompt_embeds=batch["prompt_embeds"], pooled_prompt_embeds=batch["pooled_prompt_embeds"], guidance_scale=1.0,
                                 noise_level=ad.scheduler.noise_level, compute_log_prob=True, return_kwargs=["log_prob", "dt"])
                ratio = torch.exp(out.log_prob - old_lp)
                (first_ratios if inner == 0 and start == 0 and t_idx == ad.scheduler.train_timesteps.tolist()[0] else second_ratios).append(ratio.detach().clone())
                a = torch.clamp(batch["advantage"], -5.0, 5.0)
                loss = torch.mean(torch.maximum(-a * ratio, -a * torch.clamp(ratio, 1.0 - clip, 1.0 + clip)))
                loss.backward()
                gn = torch.nn.utils.clip_grad_norm_([p for p in mod.parameters() if p.requires_grad], 1.0)
                grad_norms.append(float(gn))
                opt.step()
                opt.zero_grad()
    assert torch.equal(first_ratios[0], torch.ones_like(first_ratios[0])), first_ratios[0]          # before ANY update: exactly 1
    # (once a sample's ratio has left [1 - clip, 1 + clip] on the penalised side PPO's clipped branch wins and its gradient is exactly 0)
    assert all(np.isfinite(gn) for gn in grad_norms) and grad_norms[0] > 0 and sum(gn > 0 for gn in grad_norms) >= 4, grad_norms
    moved = torch.cat(second_ratios)
    assert float((moved - 1).abs().max()) > 1e-6            # after updates the ratio leaves 1
    assert torch.isfinite(moved).all()
    # next epoch: the rollout sees the updated weights (no explicit re-bind) -> different trajectories from the same seed
    torch.cuda.manual_seed(123)
    again = sample()
    assert not torch.equal(again[0].all_latents, samples[0].all_latents)
    print(f"GRPO epoch on the engine: first ratio == 1 exactly; max |ratio-1| after updates {float((moved - 1).abs().max()):.3e}; "
          f"grad norms {min(grad_norms):.3e} .. {max(grad_norms):.3e}")
    ad.engine.close()
