"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the Qwen-Image transformer forward and the Flow-Factory Qwen-Image rollout step
(reference src/flow_factory/models/qwen_image/qwen_image.py: `inference` :288-438, `forward`
:476-600 -- packed latents (B, h/2*w/2, 64), `timestep = t.to(latents.dtype) / 1000`, a cond and an uncond transformer call on
per-prompt text lengths, `comb = neg + g (pos - neg)` rescaled to the norm of the cond prediction (:579-587), then the same
`FlowMatchEulerDiscreteSDEScheduler.step`).

The rollout CONTROL FLOW (`rollout`, `forward_step`, `cfg_rescale_bf16`) is PINNED bit for bit against the reference's own
`QwenImageAdapter` incl. ragged prompts (tests/test_rollout_control_flow_pin.py, oracle/make_rollout_golden.py).
NETWORK BODY: PARITY UNPINNED (as oracle/mmditx_ref.py): the model body is diffusers' `QwenImageTransformer2DModel` (un-vendored third-party
dependency, constraint diffusers>=0.36.0, not installed here); it is restated from the published architecture with HF state-dict names:
  img_in Linear(64, D) / txt_norm RMSNorm(3584) / txt_in Linear(3584, D) / time_text_embed.timestep_embedder (Timesteps(256,
  flip_sin_to_cos, scale=1000) -> Linear, SiLU, Linear), 60 x QwenImageTransformerBlock (img_mod / txt_mod = SiLU + Linear(D, 6D) chunked
  (mod1 | mod2) x (shift, scale, gate); LayerNorm(no affine, 1e-6); joint attention with per-head RMSNorm q/k on both streams, complex
  RoPE (QwenEmbedRope, scale_rope: centred rows / cols, text positions after max(h/2, w/2)), text tokens FIRST in the joint sequence;
  GELU-tanh feed-forwards), AdaLayerNormContinuous + proj_out Linear(D, 64).
Padded text keys are masked out of the attention (the behaviour of diffusers >= 0.37, where `encoder_hidden_states_mask` builds the joint
mask; 0.36 ignored the mask -- the two agree whenever every prompt of a batch has the same length, e.g. one GRPO group).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass
class QwenConfig:
    in_channels: int = 64
    num_layers: int = 60
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 3584
    axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)
    scale_rope: bool = True
    time_proj_dim: int = 256
    eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


QWEN_IMAGE = QwenConfig()


def tiny_config(num_layers=2, heads=2, joint_attention_dim=128) -> QwenConfig:
    return QwenConfig(num_layers=num_layers, num_attention_heads=heads, joint_attention_dim=joint_attention_dim)


def state_dict_shapes(cfg: QwenConfig) -> Dict[str, Tuple[int, ...]]:
    D, hd, J = cfg.dim, cfg.attention_head_dim, cfg.joint_attention_dim
    out: Dict[str, Tuple[int, ...]] = {}

    def lin(n, o, i):
        out[n + ".weight"], out[n + ".bias"] = (o, i), (o,)

    lin("img_in", D, cfg.in_channels)
    out["txt_norm.weight"] = (J,)
    lin("txt_in", D, J)
    lin("time_text_embed.timestep_embedder.linear_1", D, cfg.time_proj_dim)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        lin(f"{b}.img_mod.1", 6 * D, D)
        lin(f"{b}.txt_mod.1", 6 * D, D)
        for n in ("to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out"):
            lin(f"{b}.attn.{n}", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            out[f"{b}.attn.{n}.weight"] = (hd,)
        for s in ("img_mlp", "txt_mlp"):
            lin(f"{b}.{s}.net.0.proj", 4 * D, D)
            lin(f"{b}.{s}.net.2", D, 4 * D)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", cfg.in_channels, D)
    return out


def make_synthetic_state_dict(cfg: QwenConfig, seed: int = 91, std: float = 0.03) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, s in state_dict_shapes(cfg).items():
        t = torch.randn(s, generator=g) * std
        if ("norm" in n.split(".")[-2]) and len(s) == 1:        # RMSNorm weights around 1
            t = t + 1.0
        sd[n] = t
    return sd


def timestep_embedding(t: torch.Tensor, dim: int = 256, scale: float = 1000.0) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0, scale=1000): emb = scale * (t * freqs); [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = scale * (t[:, None].float() * freqs[None])
    return torch.cat([a.cos(), a.sin()], dim=-1)


def rope_freqs(cfg: QwenConfig, hp: int, wp: int, n_text: int, theta: float = 10000.0):
    """QwenEmbedRope.forward for img_shapes = [(1, hp, wp)]: complex (Ni, 64) image and (n_text, 64) text rotations, fp32 angles."""

    def params(index, dim):
        fr = torch.outer(index.float(), 1.0 / torch.pow(torch.tensor(theta), torch.arange(0, dim, 2).float().div(dim)))
        return torch.polar(torch.ones_like(fr), fr)

    pos_index = torch.arange(4096)
    neg_index = torch.arange(4096).flip(0) * -1 - 1
    pos = [params(pos_index, d) for d in cfg.axes_dims_rope]
    neg = [params(neg_index, d) for d in cfg.axes_dims_rope]
    frame = pos[0][0:1].view(1, 1, 1, -1).expand(1, hp, wp, -1)
    if cfg.scale_rope:
        fh = torch.cat([neg[1][-(hp - hp // 2):], pos[1][:hp // 2]], dim=0)
        fw = torch.cat([neg[2][-(wp - wp // 2):], pos[2][:wp // 2]], dim=0)
        max_vid = max(hp // 2, wp // 2)
    else:
        fh, fw = pos[1][:hp], pos[2][:wp]
        max_vid = max(hp, wp)
    fh = fh.view(1, hp, 1, -1).expand(1, hp, wp, -1)
    fw = fw.view(1, 1, wp, -1).expand(1, hp, wp, -1)
    img = torch.cat([frame, fh, fw], dim=-1).reshape(hp * wp, -1)
    txt = torch.cat(pos, dim=1)[max_vid:max_vid + n_text]
    return img, txt


def apply_rope_complex(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb_qwen(use_real=False): x (B, S, H, 128) as 64 complex numbers (adjacent pairs) times freqs (S, 64)."""
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    return torch.view_as_real(xc * freqs[None, :, None, :]).flatten(3)


def _id(x):
    return x


def _rms(x, w, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def qwen_forward(sd: Dict[str, torch.Tensor], cfg: QwenConfig, hidden: torch.Tensor, timestep: torch.Tensor, enc: torch.Tensor,
                 txt_lens: Optional[Sequence[int]], hp: int, wp: int, quant: Optional[Callable] = None, return_intermediates: bool = False):
    """hidden (B, Ni, 64) packed latents; timestep (B,) = t/1000 as the adapter passes it; enc (B, Nt, 3584) zero-padded; txt_lens valid
    tokens per sample (None: all).  Returns the packed velocity (B, Ni, 64)."""
    q = quant or _id
    D, H, eps = cfg.dim, cfg.num_attention_heads, cfg.eps
    lin = lambda n, x: F.linear(q(x), q(sd[n + ".weight"]), sd[n + ".bias"])
    ln = lambda x: F.layer_norm(x, (D,), eps=eps)
    B, Nt = enc.shape[0], enc.shape[1]
    Ni = hidden.shape[1]
    inter = {}

    x = q(lin("img_in", hidden.float()))
    c = q(lin("txt_in", q(_rms(enc.float(), sd["txt_norm.weight"], eps))))
    temb = q(lin("time_text_embed.timestep_embedder.linear_2",
                 F.silu(q(lin("time_text_embed.timestep_embedder.linear_1", q(timestep_embedding(timestep.float(), cfg.time_proj_dim)))))))
    semb = q(F.silu(temb))
    img_f, txt_f = rope_freqs(cfg, hp, wp, Nt)
    mask = None
    if txt_lens is not None:
        lens = torch.as_tensor([int(v) for v in txt_lens])
        valid = torch.cat([torch.arange(Nt)[None, :] < lens[:, None], torch.ones(B, Ni, dtype=torch.bool)], dim=1)       # keys: [txt | img]
        mask = torch.zeros(B, 1, 1, Nt + Ni).masked_fill(~valid[:, None, None, :], float("-inf"))

    heads = lambda t: t.view(t.shape[0], t.shape[1], H, D // H)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        m = q(lin(f"{b}.img_mod.1", semb)).chunk(6, dim=1)       # (shift1, scale1, gate1, shift2, scale2, gate2)
        mc = q(lin(f"{b}.txt_mod.1", semb)).chunk(6, dim=1)
        xn = q(ln(x) * (1 + m[1][:, None]) + m[0][:, None])
        cn = q(ln(c) * (1 + mc[1][:, None]) + mc[0][:, None])
        qi = apply_rope_complex(_rms(heads(q(lin(f"{b}.attn.to_q", xn))), sd[f"{b}.attn.norm_q.weight"], eps), img_f)
        ki = apply_rope_complex(_rms(heads(q(lin(f"{b}.attn.to_k", xn))), sd[f"{b}.attn.norm_k.weight"], eps), img_f)
        vi = heads(lin(f"{b}.attn.to_v", xn))
        qc = apply_rope_complex(_rms(heads(q(lin(f"{b}.attn.add_q_proj", cn))), sd[f"{b}.attn.norm_added_q.weight"], eps), txt_f)
        kc = apply_rope_complex(_rms(heads(q(lin(f"{b}.attn.add_k_proj", cn))), sd[f"{b}.attn.norm_added_k.weight"], eps), txt_f)
        vc = heads(lin(f"{b}.attn.add_v_proj", cn))
        jq, jk, jv = (torch.cat([a, bb], dim=1).transpose(1, 2) for a, bb in ((qc, qi), (kc, ki), (vc, vi)))
        o = F.scaled_dot_product_attention(q(jq), q(jk), q(jv), attn_mask=mask)
        o = q(o.transpose(1, 2).reshape(B, Nt + Ni, D))
        oc, oi = o[:, :Nt], o[:, Nt:]
        x = q(x + m[2][:, None] * q(lin(f"{b}.attn.to_out.0", oi)))
        c = q(c + mc[2][:, None] * q(lin(f"{b}.attn.to_add_out", oc)))
        xn2 = q(ln(x) * (1 + m[4][:, None]) + m[3][:, None])
        x = q(x + m[5][:, None] * q(lin(f"{b}.img_mlp.net.2", q(F.gelu(lin(f"{b}.img_mlp.net.0.proj", xn2), approximate="tanh")))))
        cn2 = q(ln(c) * (1 + mc[4][:, None]) + mc[3][:, None])
        c = q(c + mc[5][:, None] * q(lin(f"{b}.txt_mlp.net.2", q(F.gelu(lin(f"{b}.txt_mlp.net.0.proj", cn2), approximate="tanh")))))
        if return_intermediates:
            inter[f"block{i}.x"], inter[f"block{i}.c"] = x, c
    mo = q(lin("norm_out.linear", semb)).chunk(2, dim=1)            # scale, shift
    xo = q(ln(x) * (1 + mo[0][:, None]) + mo[1][:, None])
    out = q(lin("proj_out", xo))
    return (out, inter) if return_intermediates else out


def forward_flops(cfg: QwenConfig, Ni: int, Nt: int) -> float:
    """Algorithmic matmul FLOPs of one forward for one sample (2 FLOP/MAC; embeddings / modulation linears excluded)."""
    D, S = cfg.dim, Ni + Nt
    blocks = cfg.num_layers * (S * 12 * D * D + 2 * S * S * D)
    emb = Ni * cfg.in_channels * D + Nt * cfg.joint_attention_dim * D + Ni * D * cfg.in_channels
    return 2.0 * (blocks + emb)


# ------------------------------------------------------------------ rollout control flow (qwen_image.py:288-438, :476-600)
def cfg_rescale_bf16(neg: torch.Tensor, pos: torch.Tensor, g: float) -> torch.Tensor:
    """qwen_image.py:579-587 on the bf16 tensors the transformer returns (every op rounds to bf16)."""
    neg, pos = neg.to(torch.bfloat16), pos.to(torch.bfloat16)
    comb = neg + g * (pos - neg)
    cond_norm = torch.norm(pos, dim=-1, keepdim=True)
    noise_norm = torch.norm(comb, dim=-1, keepdim=True)
    return comb * (cond_norm / noise_norm)


def forward_step(sd, cfg: QwenConfig, t, t_next, latents, prompt_embeds, txt_lens, hp, wp, neg_embeds=None, neg_lens=None, guidance_scale=4.0,
                 noise_level=0.0, dynamics_type="Flow-SDE", sigma_max=None, variance_noise=None, next_latents=None, compute_log_prob=True,
                 quant=None, denoiser=None):
    """qwen_image.py:476-600.  `prompt_embeds` / `neg_embeds` are padded to their batch maximum, `txt_lens` / `neg_lens` the valid lengths
    (what `_pad_batch_prompt`, :238-284, produces).  `denoiser` (tests/test_rollout_control_flow_pin.py) replaces the network AT THE
    ADAPTER'S CALL (:553-563, :567-577): it receives what the reference hands to `QwenImageTransformer2DModel`."""
    from . import scheduler_ref as S
    B = latents.shape[0]
    timestep = torch.as_tensor(t, dtype=torch.float32).reshape(-1).expand(B).to(latents.dtype)        # :497
    if denoiser is not None:
        def net(emb, lens):
            L = emb.shape[1]
            mask = (torch.arange(L)[None, :] < torch.as_tensor([int(n) for n in lens])[:, None]).long()
            return denoiser(hidden_states=latents, timestep=timestep / 1000, encoder_hidden_states=emb, encoder_hidden_states_mask=mask,
                            img_shapes=[[(1, hp, wp)]] * B, txt_seq_lens=[int(n) for n in lens]).to(torch.bfloat16)
    else:
        tm = (timestep / 1000).float()                                                                 # :534 (rounded in latents.dtype)

        def net(emb, lens):
            return qwen_forward(sd, cfg, latents.float(), tm, emb.float(), lens, hp, wp, quant=quant).to(torch.bfloat16)
    v = net(prompt_embeds, txt_lens)
    if guidance_scale > 1.0 and neg_embeds is not None:
        vn = net(neg_embeds, neg_lens)
        v = cfg_rescale_bf16(vn, v, float(guidance_scale))
    t = torch.as_tensor(t, dtype=torch.float32)
    t_next = torch.as_tensor(t_next, dtype=torch.float32)
    return S.sde_step(v, latents, t / 1000, t_next / 1000, noise_level, dynamics_type=dynamics_type, sigma_max=sigma_max,
                      variance_noise=variance_noise, next_latents=next_latents, compute_log_prob=compute_log_prob)


def rollout(sd, cfg: QwenConfig, prompt_embeds, txt_lens, neg_embeds, neg_lens, guidance_scale, init_latents, step_noise, timesteps, sigmas,
            noise_levels, hp, wp, storage_dtype=torch.bfloat16, dynamics_type="Flow-SDE", compute_log_prob=True, quant=None, denoiser=None):
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
        out = forward_step(sd, cfg, t, t_next, lat, prompt_embeds, txt_lens, hp, wp, neg_embeds, neg_lens, guidance_scale, noise_level=eta,
                           dynamics_type=dynamics_type, sigma_max=sigma_max, variance_noise=step_noise[i] if step_noise is not None else None,
                           compute_log_prob=clp, quant=quant, denoiser=denoiser)
        lat = S.cast_latents(out["next_latents"], storage_dtype)
        all_lat.append(lat)
        lps.append(out["log_prob"] if clp else torch.full((lat.shape[0],), float("nan")))
        vs.append(out["noise_pred"])
        means.append(out["next_latents_mean"])
    return dict(all_latents=torch.stack(all_lat, 0), log_probs=torch.stack(lps, 0), noise_preds=torch.stack(vs, 0),
                next_latents_means=torch.stack(means, 0))
