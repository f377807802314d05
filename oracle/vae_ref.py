"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the VAE decode at the end of a rollout:
`SD3_5Adapter.decode_latents` (reference src/flow_factory/models/stable_diffusion/sd3_5.py:161-172)
  latents / scaling_factor + shift_factor -> pipeline.vae.decode -> image_processor.postprocess('pt')
where `pipeline.vae` is diffusers' `AutoencoderKL` (SD3: latent_channels 16, block_out_channels
[128, 256, 512, 512], layers_per_block 2, norm_num_groups 32, no post_quant_conv;
scaling_factor 1.5305, shift_factor 0.0609 -- from memory of the public checkpoint config).

PARITY UNPINNED for the same reason as oracle/mmditx_ref.py: the decoder body lives in the un-vendored
third-party `diffusers`; it is restated from the published architecture (Decoder: conv_in -> mid_block
(ResnetBlock2D, single-head Attention with GroupNorm, ResnetBlock2D) -> 4 UpDecoderBlock2D (3 resnets each,
nearest-2x + conv upsampler on the first three) -> GroupNorm -> SiLU -> conv_out) with HF state-dict names.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VAEConfig:
    latent_channels: int = 16
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    eps: float = 1e-6
    scaling_factor: float = 1.5305
    shift_factor: float = 0.0609


SD3_VAE = VAEConfig()


def tiny_config() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1)


def state_dict_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    sh: Dict[str, Tuple[int, ...]] = {}

    def conv(name, co, ci, k=3):
        sh[name + ".weight"] = (co, ci, k, k)
        sh[name + ".bias"] = (co,)

    def norm(name, c):
        sh[name + ".weight"] = (c,)
        sh[name + ".bias"] = (c,)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci); conv(name + ".conv1", co, ci)
        norm(name + ".norm2", co); conv(name + ".conv2", co, co)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1)

    top = cfg.block_out_channels[-1]
    conv("decoder.conv_in", top, cfg.latent_channels)
    resnet("decoder.mid_block.resnets.0", top, top)
    norm("decoder.mid_block.attentions.0.group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sh[f"decoder.mid_block.attentions.0.{n}.weight"] = (top, top)
        sh[f"decoder.mid_block.attentions.0.{n}.bias"] = (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(cfg.block_out_channels))
    prev = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
        prev = co
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", cfg.out_channels, rev[-1])
    return sh


def make_synthetic_state_dict(cfg: VAEConfig, seed: int = 4242) -> Dict[str, torch.Tensor]:
    """Variance-preserving random init (conv / linear weights N(0, 1/fan_in), norm weights 1 + N(0, 0.1^2),
    biases N(0, 0.05^2)): activations stay O(1) through ~30 layers so every layer matters in a parity check."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in state_dict_shapes(cfg).items():
        if "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / fan_in ** 0.5
        sd[name] = t.contiguous()
    return sd


def _id(x):
    return x


def _resnet(sd, name, x, cfg, q):
    h = q(F.silu(F.group_norm(x, cfg.norm_num_groups, sd[name + ".norm1.weight"], sd[name + ".norm1.bias"], cfg.eps)))
    h = q(F.conv2d(h, q(sd[name + ".conv1.weight"]), sd[name + ".conv1.bias"], padding=1))
    h = q(F.silu(F.group_norm(h, cfg.norm_num_groups, sd[name + ".norm2.weight"], sd[name + ".norm2.bias"], cfg.eps)))
    h = q(F.conv2d(h, q(sd[name + ".conv2.weight"]), sd[name + ".conv2.bias"], padding=1))
    if (name + ".conv_shortcut.weight") in sd:
        x = q(F.conv2d(x, q(sd[name + ".conv_shortcut.weight"]), sd[name + ".conv_shortcut.bias"]))
    return q(x + h)


def _mid_attention(sd, name, x, cfg, q):
    B, C, H, W = x.shape
    h = q(F.group_norm(x, cfg.norm_num_groups, sd[name + ".group_norm.weight"], sd[name + ".group_norm.bias"], cfg.eps))
    t = h.flatten(2).transpose(1, 2)  # (B, HW, C)
    lin = lambda n, z: q(F.linear(z, q(sd[f"{name}.{n}.weight"]), sd[f"{name}.{n}.bias"]))
    qq, kk, vv = lin("to_q", t), lin("to_k", t), lin("to_v", t)
    o = q(F.scaled_dot_product_attention(qq[:, None], kk[:, None], vv[:, None])[:, 0])  # one head of dim C
    o = q(F.linear(o, q(sd[f"{name}.to_out.0.weight"]), sd[f"{name}.to_out.0.bias"]))
    return q(x + o.transpose(1, 2).reshape(B, C, H, W))


def vae_decode(sd: Dict[str, torch.Tensor], cfg: VAEConfig, latents: torch.Tensor, quant: Optional[Callable] = None,
               postprocess: bool = True) -> torch.Tensor:
    """latents (B, 16, h, w) -> images (B, 3, 8h, 8w); postprocess = (x/2 + 0.5).clamp(0, 1) ('pt' output).
    `quant` (e.g. a bf16 round trip) is applied wherever a bf16 module would materialise a tensor."""
    q = quant or _id
    z = latents.float() / cfg.scaling_factor + cfg.shift_factor
    x = q(F.conv2d(q(z), q(sd["decoder.conv_in.weight"]), sd["decoder.conv_in.bias"], padding=1))
    x = _resnet(sd, "decoder.mid_block.resnets.0", x, cfg, q)
    x = _mid_attention(sd, "decoder.mid_block.attentions.0", x, cfg, q)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x, cfg, q)
    n_up = len(cfg.block_out_channels)
    for i in range(n_up):
        for j in range(cfg.layers_per_block + 1):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, cfg, q)
        if i != n_up - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = q(F.conv2d(x, q(sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"]),
                           sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1))
    x = q(F.silu(F.group_norm(x, cfg.norm_num_groups, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], cfg.eps)))
    x = q(F.conv2d(x, q(sd["decoder.conv_out.weight"]), sd["decoder.conv_out.bias"], padding=1))
    if postprocess:
        x = q(x * 0.5 + 0.5).clamp(0, 1)   # VaeImageProcessor.denormalize on the vae-dtype tensor
    return x


def decode_flops(cfg: VAEConfig, h: int, w: int) -> float:
    """Algorithmic conv / linear / attention FLOPs per image (2 FLOP/MAC)."""
    rev = list(reversed(cfg.block_out_channels))
    top = rev[0]
    hw = h * w
    mac = hw * 9 * cfg.latent_channels * top
    mac += 2 * (2 * hw * 9 * top * top)                     # mid resnets
    mac += 4 * hw * top * top + 2 * hw * hw * top           # mid attention: q,k,v,out + QK^T + PV
    prev, res = top, hw
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            ci = prev if j == 0 else co
            mac += res * 9 * ci * co + res * 9 * co * co + (res * ci * co if ci != co else 0)
        if i != len(rev) - 1:
            res *= 4
            mac += res * 9 * co * co
        prev = co
    mac += res * 9 * rev[-1] * cfg.out_channels
    return 2.0 * mac
