"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement of the rollout control flow of the reference adapter:
`SD3_5Adapter.inference` (src/flow_factory/models/stable_diffusion/sd3_5.py:176-349) and
`SD3_5Adapter.forward` (:352-448), on top of `oracle.mmditx_ref` (denoiser) and
`oracle.scheduler_ref` (SDE step).  Precision trace follows SURVEY.md appendix:

  1. latents live in the storage dtype (models/abc.py:172-182);
  2. the network sees t rounded to that dtype (sd3_5.py:394), the scheduler the fp32 t (:438);
  3. the network output is bf16 under autocast (trainers/abc.py:72-76);
  4. CFG combine `u + g*(c-u)` is evaluated in bf16 op by op (sd3_5.py:431-433);
  5. step() upcasts to fp32, draws eps in fp32, rounds x' to the storage dtype (:362).

PINNED: with `oracle.standin.denoiser` in place of the network this loop reproduces the reference's own `SD3_5Adapter.inference()` bit for
bit (tests/test_rollout_control_flow_pin.py; fixture generator oracle/make_rollout_golden.py).

RNG: the caller passes `init_latents` and `step_noise[N]`, drawn in the reference's
order (prepare_latents, then one fp32 randn per step -- also on eta=0 steps) so that the
HIP engine and this oracle consume identical numbers.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import mmditx_ref as M
from . import scheduler_ref as S


def draw_rollout_noise(batch, channels, h, w, num_steps, latent_dtype=torch.bfloat16, generator=None, device="cpu"):
    """Reference draw order on the global generator: sd3_5.py:242 (prepare_latents, transformer
    dtype) then flow_match_euler_discrete.py:352 once per step (fp32)."""
    init = torch.randn((batch, channels, h, w), generator=generator, device=device, dtype=latent_dtype)
    noise = [torch.randn((batch, channels, h, w), generator=generator, device=device, dtype=torch.float32)
             for _ in range(num_steps)]
    return init, torch.stack(noise, 0)


def cfg_combine_bf16(uncond: torch.Tensor, text: torch.Tensor, g: float) -> torch.Tensor:
    """sd3_5.py:431-433 evaluated on bf16 tensors: each op rounds to bf16."""
    u, c = uncond.to(torch.bfloat16), text.to(torch.bfloat16)
    return u + g * (c - u)


def forward_step(
    sd, cfg: M.MMDiTConfig, t: torch.Tensor, t_next: torch.Tensor, latents: torch.Tensor,
    prompt_embeds, pooled, neg_embeds=None, neg_pooled=None, guidance_scale: float = 1.0,
    noise_level: float = 0.0, dynamics_type: str = "Flow-SDE", sigma_max: float = None,
    variance_noise=None, next_latents=None, compute_log_prob=True,
    quant: Optional[Callable] = None, denoiser: Optional[Callable] = None,
):
    """sd3_5.py:352-448.  `denoiser(latents_in, timestep_in, embeds_in, pooled_in) -> v` may replace
    the oracle network (used to feed a recorded engine output through the oracle scheduler)."""
    B = latents.shape[0]
    timestep = t.reshape(-1).expand(B).to(latents.dtype) if t.numel() == 1 else t.to(latents.dtype)
    do_cfg = neg_embeds is not None and neg_pooled is not None and guidance_scale > 1.0
    if do_cfg:
        e_in = torch.cat([neg_embeds, prompt_embeds], 0)
        p_in = torch.cat([neg_pooled, pooled], 0)
        x_in = torch.cat([latents, latents], 0)
        t_in = timestep.repeat(2)
    else:
        e_in, p_in, x_in, t_in = prompt_embeds, pooled, latents, timestep
    if denoiser is None:
        v = M.mmdit_forward(sd, cfg, x_in.float(), t_in.float(), e_in.float(), p_in.float(), quant=quant)
    else:
        v = denoiser(x_in, t_in, e_in, p_in)
    v = v.to(torch.bfloat16)  # autocast output dtype
    if do_cfg:
        vu, vt = v.chunk(2)
        v = cfg_combine_bf16(vu, vt, guidance_scale)
    sigma = t.float() / 1000
    sigma_next = t_next.float() / 1000
    out = S.sde_step(v, latents, sigma, sigma_next, noise_level, dynamics_type=dynamics_type, sigma_max=sigma_max,
                     variance_noise=variance_noise, next_latents=next_latents, compute_log_prob=compute_log_prob)
    return out


def rollout(
    sd, cfg: M.MMDiTConfig, prompt_embeds, pooled, neg_embeds, neg_pooled, guidance_scale: float,
    init_latents: torch.Tensor, step_noise: torch.Tensor, timesteps: torch.Tensor, sigmas: torch.Tensor,
    noise_levels: Sequence[float], storage_dtype: torch.dtype = torch.float16,
    dynamics_type: str = "Flow-SDE", compute_log_prob: bool = True, quant: Optional[Callable] = None,
    denoiser: Optional[Callable] = None, is_eval: bool = False,
):
    """sd3_5.py:258-304: N-step loop.  Returns dict(all_latents[N+1] (storage dtype), log_probs[N]
    (nan where not computed), noise_preds[N], next_latents_means[N]).  `denoiser` replaces the oracle network (see `forward_step`;
    tests/test_rollout_control_flow_pin.py runs this loop and the reference's own adapter on the same stand-in)."""
    N = len(timesteps)
    lat = S.cast_latents(init_latents, storage_dtype)
    all_lat = [lat]
    lps, vs, means = [], [], []
    sigma_max = float(sigmas[1])
    for i in range(N):
        t = timesteps[i]
        t_next = timesteps[i + 1] if i + 1 < N else torch.tensor(0.0)
        eta = float(noise_levels[i])
        clp = compute_log_prob and eta > 0                     # sd3_5.py:277: decided on the schedule's noise level ...
        if is_eval:
            eta = 0.0                                          # ... which step() then overrides in eval mode (flow_match_euler_discrete.py:316-317)
        out = forward_step(sd, cfg, t, t_next, lat, prompt_embeds, pooled, neg_embeds, neg_pooled, guidance_scale,
                           noise_level=eta, dynamics_type=dynamics_type, sigma_max=sigma_max,
                           variance_noise=step_noise[i], compute_log_prob=clp, quant=quant, denoiser=denoiser)
        lat = S.cast_latents(out["next_latents"], storage_dtype)
        all_lat.append(lat)
        lps.append(out["log_prob"] if clp else torch.full((lat.shape[0],), float("nan")))
        vs.append(out["noise_pred"])
        means.append(out["next_latents_mean"])
    return dict(all_latents=torch.stack(all_lat, 0), log_probs=torch.stack(lps, 0), noise_preds=torch.stack(vs, 0),
                next_latents_means=torch.stack(means, 0))
