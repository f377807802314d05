"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the denoiser on the hot path: diffusers'
`SD3Transformer2DModel` ("MMDiT-X", SD3.5-medium), called by the reference at
src/flow_factory/models/stable_diffusion/sd3_5.py:421-428.

PARITY UNPINNED: the body of this model lives in the third-party `diffusers`
package (pyproject.toml:34 `diffusers>=0.36.0`; the vendored submodule
/root/reference/diffusers is empty, no SHA recoverable) which is not installed
here, and the reference ships no tests or golden vectors for it (SURVEY.md
section 8(c)).  The published architecture is restated from the public diffusers
source (module names in the comments; state-dict keys are HF-compatible so a real
checkpoint would load):

  pos_embed (PatchEmbed: conv k2 s2 + centre-cropped pos_embed buffer)
  time_text_embed (CombinedTimestepTextProjEmbeddings)
  context_embedder (Linear 4096->D)
  transformer_blocks.{i} (JointTransformerBlock; blocks in `dual_layers` carry attn2
      and SD35AdaLayerNormZeroX; the last block is context_pre_only)
  norm_out (AdaLayerNormContinuous) ; proj_out (Linear D -> p*p*C) ; unpatchify.

`quant` lets a test insert bf16 round-trips at the points where the reference's
`torch.autocast(bf16)` run (src/flow_factory/trainers/abc.py:72-76) rounds, to separate
arithmetic-precision differences from logic errors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class MMDiTConfig:
    in_channels: int = 16
    out_channels: int = 16
    patch_size: int = 2
    num_layers: int = 24
    num_heads: int = 24
    head_dim: int = 64
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 2048
    pos_embed_max_size: int = 384
    dual_layers: Tuple[int, ...] = tuple(range(13))
    time_proj_dim: int = 256
    ff_mult: int = 4
    eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.num_heads * self.head_dim


SD35_MEDIUM = MMDiTConfig()


def tiny_config(num_layers=2, num_heads=2, dual_layers=(0,), joint_attention_dim=128, pooled_projection_dim=128,
                pos_embed_max_size=16) -> MMDiTConfig:
    return MMDiTConfig(num_layers=num_layers, num_heads=num_heads, dual_layers=tuple(dual_layers),
                       joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim,
                       pos_embed_max_size=pos_embed_max_size)


# ------------------------------------------------------------------ synthetic weights
def sincos_pos_embed(dim: int, grid: int, base_size: int) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed(embed_dim, grid, base_size=base_size) -> (1, grid*grid, dim)."""
    gh = np.arange(grid, dtype=np.float32) / (grid / base_size)
    gw = np.arange(grid, dtype=np.float32) / (grid / base_size)
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)

    def one_d(d, pos):
        omega = np.arange(d // 2, dtype=np.float64) / (d / 2.0)
        omega = 1.0 / 10000**omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one_d(dim // 2, g[0]), one_d(dim // 2, g[1])], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0)


def state_dict_shapes(cfg: MMDiTConfig) -> Dict[str, Tuple[int, ...]]:
    D, p, C = cfg.dim, cfg.patch_size, cfg.in_channels
    sh: Dict[str, Tuple[int, ...]] = {}

    def lin(name, out_f, in_f):
        sh[name + ".weight"] = (out_f, in_f)
        sh[name + ".bias"] = (out_f,)

    sh["pos_embed.proj.weight"] = (D, C, p, p)
    sh["pos_embed.proj.bias"] = (D,)
    sh["pos_embed.pos_embed"] = (1, cfg.pos_embed_max_size**2, D)
    lin("time_text_embed.timestep_embedder.linear_1", D, cfg.time_proj_dim)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, cfg.pooled_projection_dim)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    lin("context_embedder", D, cfg.joint_attention_dim)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        last = i == cfg.num_layers - 1
        dual = i in cfg.dual_layers
        lin(f"{b}.norm1.linear", (9 if dual else 6) * D, D)
        lin(f"{b}.norm1_context.linear", (2 if last else 6) * D, D)
        for n in ("to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj"):
            lin(f"{b}.attn.{n}", D, D)
        if not last:
            lin(f"{b}.attn.to_add_out", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            sh[f"{b}.attn.{n}.weight"] = (cfg.head_dim,)
        if dual:
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(f"{b}.attn2.{n}", D, D)
            for n in ("norm_q", "norm_k"):
                sh[f"{b}.attn2.{n}.weight"] = (cfg.head_dim,)
        lin(f"{b}.ff.net.0.proj", cfg.ff_mult * D, D)
        lin(f"{b}.ff.net.2", D, cfg.ff_mult * D)
        if not last:
            lin(f"{b}.ff_context.net.0.proj", cfg.ff_mult * D, D)
            lin(f"{b}.ff_context.net.2", D, cfg.ff_mult * D)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", p * p * cfg.out_channels, D)
    return sh


def make_synthetic_state_dict(cfg: MMDiTConfig, seed: int = 1234, std: float = 0.02,
                              dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
    """Non-degenerate synthetic weights: every weight and
    bias ~N(0, std^2), RMSNorm weights 1+N(0, std^2), pos_embed by the sincos formula.  No
    zero-initialised tensor anywhere (AdaLN-Zero gates and proj_out would otherwise hide whole
    blocks from a parity check)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in state_dict_shapes(cfg).items():
        if name == "pos_embed.pos_embed":
            t = sincos_pos_embed(cfg.dim, cfg.pos_embed_max_size, base_size=max(cfg.pos_embed_max_size // 6, 1))
        elif ".norm_" in name and name.endswith(".weight") and len(shape) == 1:
            t = 1.0 + std * torch.randn(shape, generator=g)
        else:
            t = std * torch.randn(shape, generator=g)
        sd[name] = t.to(dtype).contiguous()
    return sd


# ------------------------------------------------------------------ forward
def _id(x):
    return x


def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: int = 10000) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _linear(sd, name, x, q):
    return q(F.linear(q(x), q(sd[name + ".weight"].float()), sd[name + ".bias"].float()))


def _rms(x, w, eps):
    var = x.float().pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * w.float()


def _layer_norm(x, eps):
    return F.layer_norm(x.float(), (x.shape[-1],), None, None, eps)


def _attention(sd, pre, x, ctx, cfg: MMDiTConfig, q, has_add_out: bool, attn_quant=None):
    B, Ni, D = x.shape
    H, hd = cfg.num_heads, cfg.head_dim

    def heads(t):
        return t.view(B, -1, H, hd).transpose(1, 2)

    qq = heads(_linear(sd, f"{pre}.to_q", x, q))
    kk = heads(_linear(sd, f"{pre}.to_k", x, q))
    vv = heads(_linear(sd, f"{pre}.to_v", x, q))
    qq = q(_rms(qq, sd[f"{pre}.norm_q.weight"], cfg.eps))
    kk = q(_rms(kk, sd[f"{pre}.norm_k.weight"], cfg.eps))
    if ctx is not None:
        cq = heads(_linear(sd, f"{pre}.add_q_proj", ctx, q))
        ck = heads(_linear(sd, f"{pre}.add_k_proj", ctx, q))
        cv = heads(_linear(sd, f"{pre}.add_v_proj", ctx, q))
        cq = q(_rms(cq, sd[f"{pre}.norm_added_q.weight"], cfg.eps))
        ck = q(_rms(ck, sd[f"{pre}.norm_added_k.weight"], cfg.eps))
        qq = torch.cat([qq, cq], dim=2)  # image tokens first, then text
        kk = torch.cat([kk, ck], dim=2)
        vv = torch.cat([vv, cv], dim=2)
    if attn_quant is None:
        o = F.scaled_dot_product_attention(qq, kk, vv, dropout_p=0.0, is_causal=False)
    else:
        # what a flash-attention kernel rounds INSIDE the attention (the reference's bf16 run uses one): the probabilities enter the P.V
        # product in bf16 -- and, under autograd, dP comes back through the same cast in bf16.  Used by the gradient-noise isolation test.
        pm = torch.softmax((qq @ kk.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = attn_quant(pm) @ vv
    o = q(o.transpose(1, 2).reshape(B, -1, D))
    if ctx is not None:
        o, oc = o[:, :Ni], o[:, Ni:]
        oc = _linear(sd, f"{pre}.to_add_out", oc, q) if has_add_out else None
    else:
        oc = None
    o = _linear(sd, f"{pre}.to_out.0", o, q)
    return o, oc


def _ff(sd, pre, x, q):
    h = _linear(sd, f"{pre}.net.0.proj", x, _id if q is _id else q)
    h = q(F.gelu(h, approximate="tanh"))
    return _linear(sd, f"{pre}.net.2", h, q)


def mmdit_forward(
    sd: Dict[str, torch.Tensor],
    cfg: MMDiTConfig,
    hidden_states: torch.Tensor,  # (B, C, h, w)
    timestep: torch.Tensor,  # (B,) in [0, 1000] (NOT /1000: sd3_5.py:394)
    encoder_hidden_states: torch.Tensor,  # (B, Nt, joint_attention_dim)
    pooled_projections: torch.Tensor,  # (B, pooled_projection_dim)
    quant: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
    return_intermediates: bool = False,
    attn_quant: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
):
    q = quant or _id
    B, C, h, w = hidden_states.shape
    p, D = cfg.patch_size, cfg.dim
    hp, wp = h // p, w // p
    inter = {}
    # --- PatchEmbed
    x = F.conv2d(q(hidden_states.float()), q(sd["pos_embed.proj.weight"].float()), sd["pos_embed.proj.bias"].float(), stride=p)
    x = x.flatten(2).transpose(1, 2)
    m = cfg.pos_embed_max_size
    top, left = (m - hp) // 2, (m - wp) // 2
    pe = sd["pos_embed.pos_embed"].float().reshape(1, m, m, D)[:, top : top + hp, left : left + wp].reshape(1, -1, D)
    x = q(x + pe)
    # --- conditioning
    te = q(timestep_embedding(timestep.float(), cfg.time_proj_dim))
    te = _linear(sd, "time_text_embed.timestep_embedder.linear_1", te, q)
    te = _linear(sd, "time_text_embed.timestep_embedder.linear_2", q(F.silu(te)), q)
    pe_ = _linear(sd, "time_text_embed.text_embedder.linear_1", pooled_projections.float(), q)
    pe_ = _linear(sd, "time_text_embed.text_embedder.linear_2", q(F.silu(pe_)), q)
    temb = q(te + pe_)
    semb = q(F.silu(temb))
    c = _linear(sd, "context_embedder", encoder_hidden_states.float(), q)
    inter["x0"], inter["c0"], inter["temb"] = x, c, temb
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        last = i == cfg.num_layers - 1
        dual = i in cfg.dual_layers
        mod = _linear(sd, f"{b}.norm1.linear", semb, q)
        ch = mod.chunk(9 if dual else 6, dim=1)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = ch[:6]
        nx = _layer_norm(x, cfg.eps)
        xn = nx * (1 + scale_msa[:, None]) + shift_msa[:, None]
        if dual:
            shift_msa2, scale_msa2, gate_msa2 = ch[6:]
            xn2 = nx * (1 + scale_msa2[:, None]) + shift_msa2[:, None]
        cmod = _linear(sd, f"{b}.norm1_context.linear", semb, q)
        if last:
            c_scale, c_shift = cmod.chunk(2, dim=1)  # AdaLayerNormContinuous: scale first
            cn = _layer_norm(c, cfg.eps) * (1 + c_scale)[:, None] + c_shift[:, None]
        else:
            c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = cmod.chunk(6, dim=1)
            cn = _layer_norm(c, cfg.eps) * (1 + c_scale_msa[:, None]) + c_shift_msa[:, None]
        ao, aoc = _attention(sd, f"{b}.attn", xn, cn, cfg, q, has_add_out=not last, attn_quant=attn_quant)
        x = q(x + gate_msa[:, None] * ao)
        if dual:
            ao2, _ = _attention(sd, f"{b}.attn2", xn2, None, cfg, q, has_add_out=False, attn_quant=attn_quant)
            x = q(x + gate_msa2[:, None] * ao2)
        xn = _layer_norm(x, cfg.eps) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        x = q(x + gate_mlp[:, None] * _ff(sd, f"{b}.ff", xn, q))
        if not last:
            c = q(c + c_gate_msa[:, None] * aoc)
            cn = _layer_norm(c, cfg.eps) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
            c = q(c + c_gate_mlp[:, None] * _ff(sd, f"{b}.ff_context", cn, q))
        else:
            c = None
        if return_intermediates:
            inter[f"x{i + 1}"] = x
            if c is not None:
                inter[f"c{i + 1}"] = c
    omod = _linear(sd, "norm_out.linear", semb, q)
    o_scale, o_shift = omod.chunk(2, dim=1)
    x = _layer_norm(x, cfg.eps) * (1 + o_scale)[:, None] + o_shift[:, None]
    x = _linear(sd, "proj_out", x, q)
    x = x.reshape(B, hp, wp, p, p, cfg.out_channels)
    x = torch.einsum("nhwpqc->nchpwq", x).reshape(B, cfg.out_channels, hp * p, wp * p)
    if return_intermediates:
        return x, inter
    return x


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).float()


def forward_flops(cfg: MMDiTConfig, Ni: int, Nt: int) -> float:
    """algorithmic matmul FLOPs per transformer forward per sample (2 FLOP/MAC)."""
    D, Fd, L, Ld = cfg.dim, cfg.ff_mult * cfg.dim, cfg.num_layers, len(cfg.dual_layers)
    mac = (
        L * Ni * (4 * D * D + 2 * D * Fd)
        + ((L - 1) * Nt * (4 * D * D + 2 * D * Fd) + Nt * 3 * D * D)
        + Ld * Ni * 4 * D * D
        + L * 2 * (Ni + Nt) ** 2 * D
        + Ld * 2 * Ni * Ni * D
        + Nt * cfg.joint_attention_dim * D
        + 2 * Ni * (cfg.patch_size**2 * cfg.in_channels) * D
    )
    return 2.0 * mac
