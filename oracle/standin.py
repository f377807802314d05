"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

A closed-form stand-in for the denoiser network, used to PIN the rollout control-flow restatements (`oracle/rollout_ref.py`, ...) against
the reference's own adapter code: the reference's `SD3_5Adapter.inference` / `.forward` (src/flow_factory/models/stable_diffusion/
sd3_5.py:176-448) runs on CPU with this function in place of `SD3Transformer2DModel`, and the oracle's rollout runs with the same function
as its `denoiser` -- whatever differs afterwards is control flow (RNG order, dtype casts, timestep rounding, CFG order and arithmetic, what
the scheduler is handed, which positions and log-probs are kept), which is exactly what the restatement claims to reproduce.  The network
BODY stays unpinned (un-vendored diffusers); this removes everything around it from that caveat.
"""
from __future__ import annotations

import torch


def denoiser(hidden_states: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
             pooled_projections: torch.Tensor) -> torch.Tensor:
    """(B, C, h, w) latents, (B,) timestep in the latents' dtype, (B, Nt, J) / (B, P) prompt tensors -> bf16 velocity (B, C, h, w).
    Sensitive to every input (and to the batch ORDER of a CFG pair), cheap, deterministic on CPU."""
    x = hidden_states.float()
    t = timestep.float().reshape(-1, 1, 1, 1) / 1000.0
    e = encoder_hidden_states.float().mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
    p = pooled_projections.float().mean(dim=1).reshape(-1, 1, 1, 1)
    v = (0.7 * x + 0.3 * torch.roll(x, 1, dims=-1)) * torch.cos(1.3 * t) + 0.8 * torch.sin(3.0 * x) * t + 4.0 * e + 2.5 * p
    return v.to(torch.bfloat16)            # what a bf16-autocast transformer returns


def flux_denoiser(hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids) -> torch.Tensor:
    """The argument list `Flux1Adapter.forward` hands to `FluxTransformer2DModel` (reference models/flux/flux1.py:323-333): packed latents
    (B, Ni, 64), `t / 1000` in fp32, the guidance scale in the latents' dtype, pooled / T5 embeddings, zero text ids, (Ni, 3) image ids."""
    x = hidden_states.float()
    t = timestep.float().reshape(-1, 1, 1)
    g = guidance.float().reshape(-1, 1, 1)
    e = encoder_hidden_states.float().mean(dim=(1, 2)).reshape(-1, 1, 1)
    p = pooled_projections.float().mean(dim=1).reshape(-1, 1, 1)
    pos = (0.01 * img_ids.float()[:, 1] - 0.02 * img_ids.float()[:, 2]).reshape(1, -1, 1) + txt_ids.float().sum()
    v = (0.7 * x + 0.3 * torch.roll(x, 1, dims=-1)) * torch.cos(1.3 * t) + 0.8 * torch.sin(3.0 * x) * t + 0.05 * g + 4.0 * e + 2.5 * p + pos
    return v.to(torch.bfloat16)


def qwen_denoiser(hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_mask, img_shapes, txt_seq_lens) -> torch.Tensor:
    """The call of `QwenImageAdapter.forward` (reference models/qwen_image/qwen_image.py:553-563): packed latents, `timestep / 1000` in the
    latents' dtype, prompt embeddings padded to the batch maximum with their mask and lengths, one (1, h/2, w/2) shape per sample."""
    x = hidden_states.float()
    t = timestep.float().reshape(-1, 1, 1)
    m = encoder_hidden_states_mask.float()
    e = (encoder_hidden_states.float().mean(-1) * m).sum(1) / m.sum(1)                       # masked mean over the valid text tokens
    lens = torch.as_tensor([float(n) for n in txt_seq_lens]).reshape(-1, 1, 1)
    shp = float(sum(a * b * c for (a, b, c) in img_shapes[0]))
    v = (0.7 * x + 0.3 * torch.roll(x, 1, dims=-1)) * torch.cos(1.3 * t) + 0.8 * torch.sin(3.0 * x) * t + 4.0 * e.reshape(-1, 1, 1) + 0.01 * lens + 1e-4 * shp
    return v.to(torch.bfloat16)


def wan_denoiser(hidden_states, timestep, encoder_hidden_states, expert: int = 0) -> torch.Tensor:
    """The call of `Wan2_T2V_Adapter.forward` (reference models/wan/wan2_t2v.py:505-523): latents (B, 16, T, h, w) cast to the transformer's
    dtype, the scheduler's INTEGER timestep expanded to the batch, T5 embeddings.  `expert` distinguishes the two Wan2.2 transformers."""
    x = hidden_states.float()
    if timestep.ndim == 2:        # `expand_timesteps` (Wan2.2-TI2V, wan2_t2v.py:502-504): one timestep per token -- all equal in text-to-video use
        assert bool((timestep == timestep[:, :1]).all()), "per-token timesteps that differ are image conditioning: not on this path"
        timestep = timestep[:, 0]
    t = timestep.float().reshape(-1, 1, 1, 1, 1) / 1000.0
    e = encoder_hidden_states.float().mean(dim=(1, 2)).reshape(-1, 1, 1, 1, 1)
    v = (0.7 * x + 0.3 * torch.roll(x, 1, dims=2)) * torch.cos(1.3 * t) + 0.8 * torch.sin(3.0 * x) * t + 4.0 * e
    if expert:
        v = 0.5 * v + 0.1
    return v.to(torch.bfloat16)


# ---- stand-ins at the MODEL-internal level: what the network embeds after its own first arithmetic on the adapter's arguments ----------
# The engines take the values the network embeds (`t_model`, `guidance_model`), computed by the product's host code; diffusers computes the
# same values INSIDE the model (`FluxTransformer2DModel.forward`: `timestep.to(hidden_states.dtype) * 1000`, likewise guidance;
# `QwenImageTransformer2DModel`: `Timesteps(scale=1000)` on the timestep it receives).  `*_transformer_call` restates that first line and
# hands over to the model-level stand-in, so that the reference side and the engine double meet at the same quantity -- and the product's
# host arithmetic for it is what gets compared.
def flux_denoiser_model(hidden_states, t_model, guidance_model, pooled_projections, encoder_hidden_states, hp: int, wp: int) -> torch.Tensor:
    x = hidden_states.float()
    B = x.shape[0]
    t = t_model.float().reshape(-1).expand(B).reshape(-1, 1, 1) / 1000.0
    g = (guidance_model.float().reshape(-1).expand(B).reshape(-1, 1, 1) / 1000.0) if guidance_model is not None else 0.0
    e = encoder_hidden_states.float().mean(dim=(1, 2)).reshape(-1, 1, 1)
    p = pooled_projections.float().mean(dim=1).reshape(-1, 1, 1)
    rows = torch.arange(hp, dtype=torch.float32).repeat_interleave(wp)
    cols = torch.arange(wp, dtype=torch.float32).repeat(hp)
    pos = (0.01 * rows - 0.02 * cols).reshape(1, -1, 1)
    v = (0.7 * x + 0.3 * torch.roll(x, 1, dims=-1)) * torch.cos(1.3 * t) + 0.8 * torch.sin(3.0 * x) * t + 0.05 * g + 4.0 * e + 2.5 * p + pos
    return v.to(torch.bfloat16)


def flux_transformer_call(hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids) -> torch.Tensor:
    dt = hidden_states.dtype
    hp, wp = int(img_ids[:, 1].max()) + 1, int(img_ids[:, 2].max()) + 1
    return flux_denoiser_model(hidden_states, (timestep.to(dt) * 1000).float(), (guidance.to(dt) * 1000).float() if guidance is not None else None,
                               pooled_projections, encoder_hidden_states, hp, wp)


def qwen_denoiser_model(hidden_states, t_model, encoder_hidden_states, lens) -> torch.Tensor:
    """`encoder_hidden_states` padded to at least max(lens); only the first lens[b] rows of sample b count."""
    x = hidden_states.float()
    B = x.shape[0]
    t = t_model.float().reshape(-1).expand(B).reshape(-1, 1, 1) / 1000.0
    e = torch.stack([encoder_hidden_states[b, :int(n)].float().mean() for b, n in enumerate(lens)]).reshape(-1, 1, 1)
    ln = torch.as_tensor([float(n) for n in lens]).reshape(-1, 1, 1)
    v = (0.7 * x + 0.3 * torch.roll(x, 1, dims=-1)) * torch.cos(1.3 * t) + 0.8 * torch.sin(3.0 * x) * t + 4.0 * e + 0.01 * ln
    return v.to(torch.bfloat16)


def qwen_transformer_call(hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_mask, img_shapes, txt_seq_lens) -> torch.Tensor:
    return qwen_denoiser_model(hidden_states, timestep.float() * 1000.0, encoder_hidden_states, [int(n) for n in txt_seq_lens])
