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


# --------------------------------------------------------------------------------- VAE decoder
def expected_vae_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """HF `AutoencoderKL` decoder parameters (`decoder.*`) for a `mi355_flow.vae.VAEConfig`."""
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(n, co, ci, k=3):
        out[n + ".weight"], out[n + ".bias"] = (co, ci, k, k), (co,)

    def norm(n, c):
        out[n + ".weight"], out[n + ".bias"] = (c,), (c,)

    def resnet(n, ci, co):
        norm(n + ".norm1", ci); conv(n + ".conv1", co, ci)
        norm(n + ".norm2", co); conv(n + ".conv2", co, co)
        if ci != co:
            conv(n + ".conv_shortcut", co, ci, 1)

    rev = list(reversed(cfg.block_out_channels))
    top = rev[0]
    conv("decoder.conv_in", top, cfg.latent_channels)
    resnet("decoder.mid_block.resnets.0", top, top)
    norm("decoder.mid_block.attentions.0.group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"decoder.mid_block.attentions.0.{n}.weight"], out[f"decoder.mid_block.attentions.0.{n}.bias"] = (top, top), (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    prev = top
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
        prev = co
    norm("decoder.conv_norm_out", prev)
    conv("decoder.conv_out", cfg.out_channels, prev)
    return out


def synthetic_vae_state_dict(cfg, device="cuda", seed: int = 4242, dtype: torch.dtype = torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Variance-preserving random init (weights N(0, 1/fan_in), norm weights 1 + N(0, 0.1^2), biases N(0, 0.05^2)):
    activations stay O(1) through the ~30 layers, so a benchmark exercises realistic value ranges."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in expected_vae_shapes(cfg).items():
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        if "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * t
        elif name.endswith(".bias"):
            t = 0.05 * t
        else:
            t = t / math.sqrt(math.prod(shape[1:]))
        sd[name] = t.to(dtype)
    return sd


def vae_decode_flops(cfg, h: int, w: int) -> float:
    """Algorithmic conv / linear / attention FLOPs of one image decode from (h, w) latents (2 FLOP/MAC)."""
    rev = list(reversed(cfg.block_out_channels))
    top, hw = rev[0], h * w
    mac = hw * 9 * cfg.latent_channels * top + 4 * hw * 9 * top * top + 4 * hw * top * top + 2 * hw * hw * top
    prev, res = top, hw
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            ci = prev if j == 0 else co
            mac += res * 9 * ci * co + res * 9 * co * co + (res * ci * co if ci != co else 0)
        if i != len(rev) - 1:
            res *= 4
            mac += res * 9 * co * co
        prev = co
    return 2.0 * (mac + res * 9 * rev[-1] * cfg.out_channels)


# --------------------------------------------------------------------------------- FLUX.1 transformer
def expected_flux_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """HF `FluxTransformer2DModel` parameters for a `mi355_flow.flux.FluxConfig`."""
    D, hd = cfg.dim, cfg.attention_head_dim
    out: Dict[str, Tuple[int, ...]] = {}

    def lin(n, o, i):
        out[n + ".weight"], out[n + ".bias"] = (o, i), (o,)

    lin("x_embedder", D, cfg.in_channels)
    lin("context_embedder", D, cfg.joint_attention_dim)
    for e in ["timestep_embedder"] + (["guidance_embedder"] if cfg.guidance_embeds else []):
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


def synthetic_flux_state_dict(cfg, device="cuda", seed: int = 2468, std: float = 0.02, dtype: torch.dtype = torch.bfloat16):
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in expected_flux_shapes(cfg).items():
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std
        if ".norm_" in name and len(shape) == 1:
            t = t + 1.0
        sd[name] = t.to(dtype)
    return sd


def flux_forward_flops(cfg, Ni: int, Nt: int) -> float:
    """Algorithmic matmul FLOPs of one FLUX forward for one sample (2 FLOP/MAC; conditioning MLPs / modulation linears excluded)."""
    D, S = cfg.dim, Ni + Nt
    dbl = cfg.num_layers * (S * 12 * D * D + 2 * S * S * D)
    sgl = cfg.num_single_layers * (S * 12 * D * D + 2 * S * S * D)
    emb = Ni * cfg.in_channels * D + Nt * cfg.joint_attention_dim * D + Ni * D * cfg.in_channels
    return 2.0 * (dbl + sgl + emb)


# --------------------------------------------------------------------------------- Wan2.1 transformer
def expected_wan_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """HF `WanTransformer3DModel` parameters for a `mi355_flow.wan.WanConfig`."""
    D = cfg.dim
    out: Dict[str, Tuple[int, ...]] = {}

    def lin(n, o, i):
        out[n + ".weight"], out[n + ".bias"] = (o, i), (o,)

    out["patch_embedding.weight"], out["patch_embedding.bias"] = (D, cfg.in_channels) + tuple(cfg.patch_size), (D,)
    lin("condition_embedder.time_embedder.linear_1", D, cfg.freq_dim)
    lin("condition_embedder.time_embedder.linear_2", D, D)
    lin("condition_embedder.time_proj", 6 * D, D)
    lin("condition_embedder.text_embedder.linear_1", D, cfg.text_dim)
    lin("condition_embedder.text_embedder.linear_2", D, D)
    for i in range(cfg.num_layers):
        b = f"blocks.{i}"
        out[f"{b}.scale_shift_table"] = (1, 6, D)
        for a in ("attn1", "attn2"):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(f"{b}.{a}.{n}", D, D)
            out[f"{b}.{a}.norm_q.weight"] = (D,)
            out[f"{b}.{a}.norm_k.weight"] = (D,)
        out[f"{b}.norm2.weight"], out[f"{b}.norm2.bias"] = (D,), (D,)
        lin(f"{b}.ffn.net.0.proj", cfg.ffn_dim, D)
        lin(f"{b}.ffn.net.2", D, cfg.ffn_dim)
    out["scale_shift_table"] = (1, 2, D)
    lin("proj_out", cfg.out_channels * math.prod(cfg.patch_size), D)
    return out


def synthetic_wan_state_dict(cfg, device="cuda", seed: int = 1357, std: float = 0.02, dtype: torch.dtype = torch.bfloat16):
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in expected_wan_shapes(cfg).items():
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std
        if ".norm_q." in name or ".norm_k." in name or name.endswith("norm2.weight"):
            t = t + 1.0
        if "scale_shift_table" in name:
            t = t * (0.3 / std)
        sd[name] = t.to(dtype)
    return sd


def wan_forward_flops(cfg, S: int, Nt: int) -> float:
    """Algorithmic matmul FLOPs of one Wan forward for one sample (2 FLOP/MAC; conditioning MLPs excluded)."""
    D, Fd = cfg.dim, cfg.ffn_dim
    per_layer = S * (4 * D * D) + 2 * S * S * D + S * 2 * D * D + Nt * 2 * D * D + 2 * S * Nt * D + S * 2 * D * Fd
    emb = S * cfg.in_channels * 4 * D + S * D * cfg.out_channels * 4 + Nt * cfg.text_dim * D + Nt * D * D
    return 2.0 * (cfg.num_layers * per_layer + emb)


# --------------------------------------------------------------------------------- torch module with HF parameter names
def module_from_state_dict(state_dict: Dict[str, torch.Tensor], buffers: Tuple[str, ...] = ("pos_embed.pos_embed",)) -> torch.nn.Module:
    """A parameter-only `nn.Module` tree whose attribute paths / `named_parameters()` are exactly the keys of `state_dict`
    (e.g. `transformer_blocks.3.attn.to_q.weight`).  It has no forward: the engine is the forward.  Bound to a standalone adapter
    (`SD3_5NativeAdapter(module, ...)`) it gives optimizers, DDP, LoRA wrappers and checkpoints something to hold on to, and
    grad-mode `adapter.forward()` differentiates w.r.t. whichever of its parameters have `requires_grad` (mi355_flow/autograd.py)."""
    root = torch.nn.Module()
    for name, value in state_dict.items():
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, torch.nn.Module())
            mod = getattr(mod, p)
        if name in buffers:
            mod.register_buffer(parts[-1], value)
        else:
            mod.register_parameter(parts[-1], torch.nn.Parameter(value, requires_grad=False))
    return root
