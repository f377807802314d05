"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the FLUX.1 transformer forward and the Flow-Factory FLUX.1 rollout step
(reference src/flow_factory/models/flux/flux1.py:151-346: `inference` :151-289, `forward`
:294-346 -- packed latents (B, h/2*w/2, 64), `timestep = t / 1000`, embedded guidance (no CFG), `txt_ids` zeros,
`img_ids` from `prepare_latents`, then the same `FlowMatchEulerDiscreteSDEScheduler.step`).

The rollout CONTROL FLOW (`rollout`, `forward_step`) is PINNED bit for bit against the reference's own `Flux1Adapter`
(tests/test_rollout_control_flow_pin.py, oracle/make_rollout_golden.py).
NETWORK BODY: PARITY UNPINNED (as oracle/mmditx_ref.py): the model body is diffusers' `FluxTransformer2DModel` (un-vendored third-party
dependency, not installed here); it is restated from the published architecture with HF state-dict names:
  x_embedder / context_embedder / time_text_embed.{timestep,guidance,text}_embedder,
  19 x FluxTransformerBlock (AdaLayerNormZero x2, joint attention with per-head RMSNorm q/k and RoPE over
      cat(txt_ids, img_ids), context tokens FIRST in the joint sequence, GELU-tanh feed-forwards),
  38 x FluxSingleTransformerBlock (AdaLayerNormZeroSingle, parallel attention + MLP, proj_out over cat([attn, mlp])),
  AdaLayerNormContinuous + proj_out.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class FluxConfig:
    in_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)
    time_proj_dim: int = 256
    eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


FLUX1_DEV = FluxConfig()


def tiny_config(num_layers=2, num_single_layers=2, heads=2, joint_attention_dim=128, pooled_projection_dim=64) -> FluxConfig:
    return FluxConfig(num_layers=num_layers, num_single_layers=num_single_layers, num_attention_heads=heads,
                      joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim)


def state_dict_shapes(cfg: FluxConfig) -> Dict[str, Tuple[int, ...]]:
    D, hd = cfg.dim, cfg.attention_head_dim
    out: Dict[str, Tuple[int, ...]] = {}

    def lin(n, o, i):
        out[n + ".weight"], out[n + ".bias"] = (o, i), (o,)

    lin("x_embedder", D, cfg.in_channels)
    lin("context_embedder", D, cfg.joint_attention_dim)
    emb = ["timestep_embedder"] + (["guidance_embedder"] if cfg.guidance_embeds else [])
    for e in emb:
        lin(f"time_text_embed.{e}.linear_1", D, cfg.time_proj_dim)
        lin(f"time_text_embed.{e}.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, cfg.pooled_projection_dim)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        lin(f"{b}.norm1.linear", 6 * D, D)
        lin(f"{b}.norm1_context.linear", 6 * D, D)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(f"{b}.attn.{n}", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            out[f"{b}.attn.{n}.weight"] = (hd,)
        lin(f"{b}.ff.net.0.proj", 4 * D, D); lin(f"{b}.ff.net.2", D, 4 * D)
        lin(f"{b}.ff_context.net.0.proj", 4 * D, D); lin(f"{b}.ff_context.net.2", D, 4 * D)
    for i in range(cfg.num_single_layers):
        b = f"single_transformer_blocks.{i}"
        lin(f"{b}.norm.linear", 3 * D, D)
        lin(f"{b}.proj_mlp", 4 * D, D)
        lin(f"{b}.proj_out", D, 5 * D)
        for n in ("to_q", "to_k", "to_v"):
            lin(f"{b}.attn.{n}", D, D)
        for n in ("norm_q", "norm_k"):
            out[f"{b}.attn.{n}.weight"] = (hd,)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", cfg.in_channels, D)
    return out


def make_synthetic_state_dict(cfg: FluxConfig, seed: int = 77, std: float = 0.03) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, s in state_dict_shapes(cfg).items():
        t = torch.randn(s, generator=g) * std
        if ".norm_" in n and len(s) == 1:
            t = t + 1.0
        sd[n] = t
    return sd


def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([a.cos(), a.sin()], dim=-1)


def rope_cos_sin(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0):
    """FluxPosEmbed: ids (S, 3) -> cos, sin (S, sum(axes_dim)); pair (2j, 2j+1) of an axis shares the angle pos*theta^(-2j/dim)."""
    cos, sin = [], []
    pos = ids.double()
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = torch.outer(pos[:, i], freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, dim=-1), torch.cat(sin, dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x (B, H, S, D): out = x*cos + rotate(x)*sin with rotate((a, b)) = (-b, a) on adjacent pairs."""
    xr = x.reshape(*x.shape[:-1], -1, 2)
    rot = torch.stack([-xr[..., 1], xr[..., 0]], dim=-1).flatten(3)
    return x * cos[None, None] + rot * sin[None, None]


def prepare_img_ids(hp: int, wp: int) -> torch.Tensor:
    """FluxPipeline._prepare_latent_image_ids: (hp*wp, 3) = [0, row, col]."""
    ids = torch.zeros(hp, wp, 3)
    ids[..., 1] += torch.arange(hp)[:, None]
    ids[..., 2] += torch.arange(wp)[None, :]
    return ids.reshape(hp * wp, 3)


def pack_latents(lat: torch.Tensor) -> torch.Tensor:
    """FluxPipeline._pack_latents: (B, C, h, w) -> (B, h/2*w/2, 4C), feature = c*4 + ph*2 + pw."""
    B, C, h, w = lat.shape
    return lat.view(B, C, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, (h // 2) * (w // 2), C * 4)


def unpack_latents(x: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """inverse of pack_latents: (B, h/2*w/2, 4C) -> (B, C, h, w)."""
    B, _, ch = x.shape
    return x.view(B, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(B, ch // 4, h, w)


def _id(x):
    return x


def _rms(x, w, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def _heads(x, H):
    B, S, D = x.shape
    return x.view(B, S, H, D // H).transpose(1, 2)


def flux_forward(sd: Dict[str, torch.Tensor], cfg: FluxConfig, hidden: torch.Tensor, timestep: torch.Tensor, guidance: torch.Tensor,
                 pooled: torch.Tensor, enc: torch.Tensor, img_ids: torch.Tensor, txt_ids: Optional[torch.Tensor] = None,
                 quant: Optional[Callable] = None, return_intermediates: bool = False, premultiplied: bool = False):
    """hidden (B, Ni, 64) packed latents; timestep = t/1000 and guidance as the adapter passes them (the model multiplies
    both by 1000; `premultiplied` = they already carry the x1000, e.g. rounded in the latent dtype); pooled (B, 768); enc (B, Nt, 4096); img_ids (Ni, 3).  Returns the packed velocity (B, Ni, 64)."""
    q = quant or _id
    k1000 = 1.0 if premultiplied else 1000.0
    D, H, eps = cfg.dim, cfg.num_attention_heads, cfg.eps
    lin = lambda n, x: F.linear(q(x), q(sd[n + ".weight"]), sd[n + ".bias"])
    ln = lambda x: F.layer_norm(x, (D,), eps=eps)
    Nt = enc.shape[1]
    inter = {}

    x = q(lin("x_embedder", hidden.float()))
    c = q(lin("context_embedder", enc.float()))
    t_emb = lin("time_text_embed.timestep_embedder.linear_2", F.silu(q(lin("time_text_embed.timestep_embedder.linear_1",
                                                                          q(timestep_embedding(timestep.float() * k1000, cfg.time_proj_dim))))))
    temb = q(t_emb)
    if cfg.guidance_embeds:
        g_emb = lin("time_text_embed.guidance_embedder.linear_2", F.silu(q(lin("time_text_embed.guidance_embedder.linear_1",
                                                                              q(timestep_embedding(guidance.float() * k1000, cfg.time_proj_dim))))))
        temb = q(temb + q(g_emb))
    p_emb = lin("time_text_embed.text_embedder.linear_2", F.silu(q(lin("time_text_embed.text_embedder.linear_1", pooled.float()))))
    temb = q(temb + q(p_emb))
    semb = q(F.silu(temb))
    if txt_ids is None:
        txt_ids = torch.zeros(Nt, 3)
    cos, sin = rope_cos_sin(torch.cat([txt_ids.float(), img_ids.float()], dim=0), cfg.axes_dims_rope)

    def attention(qq, kk, vv):
        qq, kk = apply_rope(qq, cos, sin), apply_rope(kk, cos, sin)
        o = F.scaled_dot_product_attention(q(qq), q(kk), q(vv))
        return q(o.transpose(1, 2).reshape(o.shape[0], o.shape[2], D))

    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        m = q(lin(f"{b}.norm1.linear", semb)).chunk(6, dim=1)      # shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        mc = q(lin(f"{b}.norm1_context.linear", semb)).chunk(6, dim=1)
        xn = q(ln(x) * (1 + m[1][:, None]) + m[0][:, None])
        cn = q(ln(c) * (1 + mc[1][:, None]) + mc[0][:, None])
        qi = _rms(_heads(q(lin(f"{b}.attn.to_q", xn)), H), sd[f"{b}.attn.norm_q.weight"], eps)
        ki = _rms(_heads(q(lin(f"{b}.attn.to_k", xn)), H), sd[f"{b}.attn.norm_k.weight"], eps)
        vi = _heads(lin(f"{b}.attn.to_v", xn), H)
        qc = _rms(_heads(q(lin(f"{b}.attn.add_q_proj", cn)), H), sd[f"{b}.attn.norm_added_q.weight"], eps)
        kc = _rms(_heads(q(lin(f"{b}.attn.add_k_proj", cn)), H), sd[f"{b}.attn.norm_added_k.weight"], eps)
        vc = _heads(lin(f"{b}.attn.add_v_proj", cn), H)
        o = attention(torch.cat([qc, qi], dim=2), torch.cat([kc, ki], dim=2), torch.cat([vc, vi], dim=2))
        oc, oi = o[:, :Nt], o[:, Nt:]
        x = q(x + m[2][:, None] * q(lin(f"{b}.attn.to_out.0", oi)))
        xn2 = q(ln(x) * (1 + m[4][:, None]) + m[3][:, None])
        x = q(x + m[5][:, None] * q(lin(f"{b}.ff.net.2", q(F.gelu(lin(f"{b}.ff.net.0.proj", xn2), approximate="tanh")))))
        c = q(c + mc[2][:, None] * q(lin(f"{b}.attn.to_add_out", oc)))
        cn2 = q(ln(c) * (1 + mc[4][:, None]) + mc[3][:, None])
        c = q(c + mc[5][:, None] * q(lin(f"{b}.ff_context.net.2", q(F.gelu(lin(f"{b}.ff_context.net.0.proj", cn2), approximate="tanh")))))
        if return_intermediates:
            inter[f"double{i}.x"], inter[f"double{i}.c"] = x, c

    y = torch.cat([c, x], dim=1)
    for i in range(cfg.num_single_layers):
        b = f"single_transformer_blocks.{i}"
        m = q(lin(f"{b}.norm.linear", semb)).chunk(3, dim=1)        # shift, scale, gate
        yn = q(ln(y) * (1 + m[1][:, None]) + m[0][:, None])
        mlp = q(F.gelu(lin(f"{b}.proj_mlp", yn), approximate="tanh"))
        qq = _rms(_heads(q(lin(f"{b}.attn.to_q", yn)), H), sd[f"{b}.attn.norm_q.weight"], eps)
        kk = _rms(_heads(q(lin(f"{b}.attn.to_k", yn)), H), sd[f"{b}.attn.norm_k.weight"], eps)
        vv = _heads(lin(f"{b}.attn.to_v", yn), H)
        o = attention(qq, kk, vv)
        y = q(y + m[2][:, None] * q(lin(f"{b}.proj_out", torch.cat([o, mlp], dim=2))))
        if return_intermediates:
            inter[f"single{i}.y"] = y
    x = y[:, Nt:]
    mo = q(lin("norm_out.linear", semb)).chunk(2, dim=1)            # scale, shift
    xo = q(ln(x) * (1 + mo[0][:, None]) + mo[1][:, None])
    out = q(lin("proj_out", xo))
    return (out, inter) if return_intermediates else out


def forward_flops(cfg: FluxConfig, Ni: int, Nt: int) -> float:
    """Algorithmic matmul FLOPs of one forward for one sample (2 FLOP/MAC; embeddings / modulation linears excluded)."""
    D, S = cfg.dim, Ni + Nt
    dbl = cfg.num_layers * (S * (4 * D * D + 8 * D * D) + 2 * S * S * D)            # qkv+out, ff (4D each way), QK^T + PV
    sgl = cfg.num_single_layers * (S * (3 * D * D + 4 * D * D + 5 * D * D) + 2 * S * S * D)
    emb = Ni * cfg.in_channels * D + Nt * cfg.joint_attention_dim * D + Ni * D * cfg.in_channels
    return 2.0 * (dbl + sgl + emb)


# ------------------------------------------------------------------ rollout control flow (flux1.py:151-289, :294-346)
def model_scalar(value, dtype: torch.dtype) -> torch.Tensor:
    """FluxTransformer2DModel.forward: `x.to(hidden_states.dtype) * 1000` (x = t/1000 or guidance; hidden_states = latents in the
    storage dtype), returned as the fp32 VALUE the sinusoidal projection sees."""
    return (torch.as_tensor(value, dtype=torch.float32).to(dtype) * 1000).float()


def forward_step(sd, cfg: FluxConfig, t, t_next, latents, prompt_embeds, pooled, img_ids, guidance_scale=3.5, noise_level=0.0,
                 dynamics_type="Flow-SDE", sigma_max=None, variance_noise=None, next_latents=None, compute_log_prob=True, quant=None,
                 denoiser=None):
    """flux1.py:294-346.  `denoiser` (tests/test_rollout_control_flow_pin.py) replaces the network AT THE ADAPTER'S CALL (flux1.py:323-333):
    it receives exactly what the reference hands to `FluxTransformer2DModel`."""
    from . import scheduler_ref as S
    B = latents.shape[0]
    if denoiser is not None:
        v = denoiser(hidden_states=latents, timestep=torch.as_tensor(t, dtype=torch.float32).expand(B) / 1000,
                     guidance=torch.as_tensor(guidance_scale, dtype=latents.dtype).expand(B), pooled_projections=pooled,
                     encoder_hidden_states=prompt_embeds, txt_ids=torch.zeros(prompt_embeds.shape[1], 3).to(dtype=latents.dtype), img_ids=img_ids)
    else:
        tm = model_scalar(t.float() / 1000, latents.dtype).reshape(-1).expand(B)
        gm = model_scalar(torch.full((B,), float(guidance_scale)).to(latents.dtype).float(), latents.dtype)
        v = flux_forward(sd, cfg, latents.float(), tm, gm, pooled.float(), prompt_embeds.float(), img_ids, quant=quant, premultiplied=True)
    v = v.to(torch.bfloat16)
    return S.sde_step(v, latents, t.float() / 1000, t_next.float() / 1000, noise_level, dynamics_type=dynamics_type, sigma_max=sigma_max,
                      variance_noise=variance_noise, next_latents=next_latents, compute_log_prob=compute_log_prob)


def rollout(sd, cfg: FluxConfig, prompt_embeds, pooled, guidance_scale, init_latents, step_noise, timesteps, sigmas, noise_levels,
            img_ids, storage_dtype=torch.float16, dynamics_type="Flow-SDE", compute_log_prob=True, quant=None, denoiser=None):
    from . import scheduler_ref as S
    N = len(timesteps)
    lat = S.cast_latents(init_latents, storage_dtype)
    all_lat, lps, vs, means = [lat], [], [], []
    sigma_max = float(sigmas[1])
    for i in range(N):
        t = timesteps[i]
        t_next = timesteps[i + 1] if i + 1 < N else torch.tensor(0.0)
        eta = float(noise_levels[i])
        clp = compute_log_prob and eta > 0
        out = forward_step(sd, cfg, t, t_next, lat, prompt_embeds, pooled, img_ids, guidance_scale, noise_level=eta,
                           dynamics_type=dynamics_type, sigma_max=sigma_max, variance_noise=step_noise[i], compute_log_prob=clp, quant=quant,
                           denoiser=denoiser)
        lat = S.cast_latents(out["next_latents"], storage_dtype)
        all_lat.append(lat)
        lps.append(out["log_prob"] if clp else torch.full((lat.shape[0],), float("nan")))
        vs.append(out["noise_pred"])
        means.append(out["next_latents_mean"])
    return dict(all_latents=torch.stack(all_lat, 0), log_probs=torch.stack(lps, 0), noise_preds=torch.stack(vs, 0),
                next_latents_means=torch.stack(means, 0))
