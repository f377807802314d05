"""Parameter table of the SD3.5 MMDiT-X transformer (HF state-dict names -> shapes) and a synthetic
initialiser for benchmarks (there are no checkpoints in this environment).  Real weights bind
through the same names: `Engine.bind_state_dict(pipeline.transformer.state_dict())`."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

from .engine import TransformerConfig


def expected_shapes(cfg: TransformerConfig) -> Dict[str, Tuple[int, ...]]:
    D, p, C, hd = cfg.dim, cfg.patch_size, cfg.in_channels, cfg.head_dim
    F = cfg.ff_mult * D
    out: Dict[str, Tuple[int, ...]] = {}

    def lin(n, o, i):
        out[n + ".weight"], out[n + ".bias"] = (o, i), (o,)

    out["pos_embed.proj.weight"], out["pos_embed.proj.bias"] = (D, C, p, p), (D,)
    out["pos_embed.pos_embed"] = (1, cfg.pos_embed_max_size ** 2, D)
    lin("time_text_embed.timestep_embedder.linear_1", D, cfg.time_proj_dim)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, cfg.pooled_projection_dim)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    lin("context_embedder", D, cfg.joint_attention_dim)
    for i in range(cfg.num_layers):
        b, last, dual = f"transformer_blocks.{i}", i == cfg.num_layers - 1, i in cfg.dual_layers
        lin(f"{b}.norm1.linear", (9 if dual else 6) * D, D)
        lin(f"{b}.norm1_context.linear", (2 if last else 6) * D, D)
        for n in ("to_q", "to_k", "to_v", "to_out.0", "add_q_proj", "add_k_proj", "add_v_proj"):
            lin(f"{b}.attn.{n}", D, D)
        if not last:
            lin(f"{b}.attn.to_add_out", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            out[f"{b}.attn.{n}.weight"] = (hd,)
        if dual:
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(f"{b}.attn2.{n}", D, D)
            for n in ("norm_q", "norm_k"):
                out[f"{b}.attn2.{n}.weight"] = (hd,)
        lin(f"{b}.ff.net.0.proj", F, D)
        lin(f"{b}.ff.net.2", D, F)
        if not last:
            lin(f"{b}.ff_context.net.0.proj", F, D)
            lin(f"{b}.ff_context.net.2", D, F)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", p * p * cfg.out_channels, D)
    return out


def synthetic_state_dict(cfg: TransformerConfig, device="cuda", seed: int = 1234, std: float = 0.02,
                         dtype: torch.dtype = torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Random-init weights of the named architecture: N(0, std^2), RMSNorm weights 1 + N(0, std^2);
    nothing zero-initialised (AdaLN-Zero gates / proj_out stay live)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in expected_shapes(cfg).items():
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std
        if ".norm_" in name and len(shape) == 1:
            t = t + 1.0
        if name == "pos_embed.pos_embed":
            t = t * (0.5 / std)  # O(1) like the sincos table
        sd[name] = t.to(dtype)
    return sd
