"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the causal 3-D video VAE decode that closes a Wan rollout and (with one frame) a Qwen-Image rollout
(reference src/flow_factory/models/wan/wan2_t2v.py:215-230 `decode_latents`: `latents / (1/std) + mean` ->
`pipeline.vae.decode` -> `video_processor.postprocess_video`; src/flow_factory/models/qwen_image/qwen_image.py:197-213: the same
decoder on a single frame, `[:, :, 0]`).

PARITY UNPINNED (as oracle/vae_ref.py): the decoder body is diffusers' `AutoencoderKLWan` / `AutoencoderKLQwenImage` (un-vendored
third-party dependency, not installed here), restated from the published architecture with HF state-dict names:
  post_quant_conv (1x1x1) -> decoder.conv_in (causal 3x3x3) -> mid_block (resnet, per-frame single-head attention, resnet) ->
  4 up_blocks of 3 residual blocks (RMS-norm over channels, SiLU, causal 3x3x3 convs, 1x1x1 shortcut when the width changes) with
  `upsample3d` (time_conv (3,1,1) doubling the frames, then nearest 2x + Conv2d halving the channels), `upsample3d`, `upsample2d`, none ->
  norm_out, SiLU, conv_out (causal 3x3x3) -> clamp(-1, 1).
Two formulations are given and tested against each other (tests/test_wan_vae_oracle.py):
  * `decode_chunked`: the published frame-by-frame algorithm with its feature cache (`CACHE_T = 2` frames per causal conv, the "Rep"
    marker that makes the first latent frame skip the temporal upsampling);
  * `decode_full`: the whole-sequence form the engine implements -- every causal conv over the full frame axis with two zero frames
    in front; the temporal upsampler keeps frame 0 and runs its time_conv over frames 1.. as a sequence of their own.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

CACHE_T = 2


@dataclass
class WanVAEConfig:
    z_dim: int = 16
    base_dim: int = 96
    dim_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    temperal_downsample: Tuple[bool, ...] = (False, True, True)
    out_channels: int = 3
    latents_mean: Tuple[float, ...] = (-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517,
                                       -0.3632, -0.1922, -0.9497, 0.2503, -0.2921)
    latents_std: Tuple[float, ...] = (2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579,
                                      1.6382, 1.1253, 2.8251, 1.9160)

    @property
    def dims(self) -> List[int]:
        return [self.base_dim * u for u in [self.dim_mult[-1]] + list(self.dim_mult[::-1])]

    @property
    def temperal_upsample(self) -> List[bool]:
        return list(self.temperal_downsample[::-1])

    def up_plan(self):
        """[(in_dim, out_dim, mode)] of the up blocks; mode in ('upsample3d', 'upsample2d', None)."""
        d = self.dims
        out = []
        for i, (a, b) in enumerate(zip(d[:-1], d[1:])):
            if i > 0:
                a = a // 2
            mode = None
            if i != len(self.dim_mult) - 1:
                mode = "upsample3d" if self.temperal_upsample[i] else "upsample2d"
            out.append((a, b, mode))
        return out


WAN21 = WanVAEConfig()


def tiny_config() -> WanVAEConfig:
    return WanVAEConfig(base_dim=32, dim_mult=(1, 2, 2, 2))


def state_dict_shapes(cfg: WanVAEConfig) -> Dict[str, Tuple[int, ...]]:
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(n, co, ci, k):
        out[n + ".weight"], out[n + ".bias"] = (co, ci) + tuple(k), (co,)

    def res(n, ci, co):
        out[n + ".norm1.gamma"] = (ci, 1, 1, 1)
        conv(n + ".conv1", co, ci, (3, 3, 3))
        out[n + ".norm2.gamma"] = (co, 1, 1, 1)
        conv(n + ".conv2", co, co, (3, 3, 3))
        if ci != co:
            conv(n + ".conv_shortcut", co, ci, (1, 1, 1))

    conv("post_quant_conv", cfg.z_dim, cfg.z_dim, (1, 1, 1))
    top = cfg.dims[0]
    conv("decoder.conv_in", top, cfg.z_dim, (3, 3, 3))
    res("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    out[a + ".norm.gamma"] = (top, 1, 1)
    conv(a + ".to_qkv", 3 * top, top, (1, 1))
    conv(a + ".proj", top, top, (1, 1))
    res("decoder.mid_block.resnets.1", top, top)
    for i, (ci, co, mode) in enumerate(cfg.up_plan()):
        for j in range(cfg.num_res_blocks + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", ci if j == 0 else co, co)
        if mode is not None:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.resample.1", co // 2, co, (3, 3))
            if mode == "upsample3d":
                conv(f"decoder.up_blocks.{i}.upsamplers.0.time_conv", 2 * co, co, (3, 1, 1))
    last = cfg.dims[-1]
    out["decoder.norm_out.gamma"] = (last, 1, 1, 1)
    conv("decoder.conv_out", cfg.out_channels, last, (3, 3, 3))
    return out


def make_synthetic_state_dict(cfg: WanVAEConfig, seed: int = 21) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, s in state_dict_shapes(cfg).items():
        if n.endswith(".gamma"):
            sd[n] = 1.0 + 0.1 * torch.randn(s, generator=g)
        elif n.endswith(".bias"):
            sd[n] = 0.02 * torch.randn(s, generator=g)
        else:
            fan_in = 1
            for d in s[1:]:
                fan_in *= d
            sd[n] = torch.randn(s, generator=g) * (1.0 / fan_in) ** 0.5
    return sd


# ------------------------------------------------------------------------------------------------ building blocks
def rms_norm(x: torch.Tensor, gamma: torch.Tensor) -> torch.Tensor:
    """WanRMS_norm(channel_first): F.normalize(x, dim=1) * sqrt(C) * gamma."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * gamma


def causal_conv3d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, cache: Optional[torch.Tensor] = None) -> torch.Tensor:
    """WanCausalConv3d.forward: spatial zero padding k//2, temporal padding 2*(kt//2) frames in FRONT, shortened by the cached frames."""
    kt, kh, kw = w.shape[2:]
    pt = 2 * (kt // 2)
    if cache is not None and pt > 0:
        x = torch.cat([cache, x], dim=2)
        pt -= cache.shape[2]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, pt, 0))
    return F.conv3d(x, w, b)


def _per_frame(x, fn):
    B, C, T, H, W = x.shape
    y = fn(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W))
    return y.reshape(B, T, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def attention_block(sd, n, x):
    """WanAttentionBlock: per-frame single-head self-attention over the h*w positions, head dim C."""
    def fn(f):
        h = rms_norm(f, sd[n + ".norm.gamma"])
        qkv = F.conv2d(h, sd[n + ".to_qkv.weight"], sd[n + ".to_qkv.bias"])
        BT, C3, H, W = qkv.shape
        q, k, v = qkv.reshape(BT, 1, C3, H * W).permute(0, 1, 3, 2).chunk(3, dim=-1)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.squeeze(1).permute(0, 2, 1).reshape(BT, C3 // 3, H, W)
        return F.conv2d(o, sd[n + ".proj.weight"], sd[n + ".proj.bias"])
    return x + _per_frame(x, fn)


def spatial_up_conv(sd, n, x):
    """resample = Sequential(WanUpsample(2x nearest-exact), Conv2d(dim, dim // 2, 3, padding=1)) on every frame."""
    return _per_frame(x, lambda f: F.conv2d(F.interpolate(f.float(), scale_factor=2.0, mode="nearest-exact"), sd[n + ".resample.1.weight"],
                                            sd[n + ".resample.1.bias"], padding=1))


def _interleave(y):
    """(B, 2C, T, H, W) -> (B, C, 2T, H, W): frame 2t + s takes channels [s*C, (s+1)*C)."""
    B, C2, T, H, W = y.shape
    y = y.reshape(B, 2, C2 // 2, T, H, W)
    return torch.stack((y[:, 0], y[:, 1]), 3).reshape(B, C2 // 2, 2 * T, H, W)


# ------------------------------------------------------------------------------------------------ whole-sequence form
def decode_full(sd: Dict[str, torch.Tensor], cfg: WanVAEConfig, z: torch.Tensor, return_stages: bool = False):
    """z (B, 16, T, h, w) DE-normalised latents -> video (B, 3, 1 + 4 (T - 1), 8h, 8w) in [-1, 1]."""
    conv = lambda n, x: causal_conv3d(x, sd[n + ".weight"], sd[n + ".bias"])
    stages = {}

    def res(n, x):
        h = conv(n + ".conv_shortcut", x) if (n + ".conv_shortcut.weight") in sd else x
        y = conv(n + ".conv1", F.silu(rms_norm(x, sd[n + ".norm1.gamma"])))
        y = conv(n + ".conv2", F.silu(rms_norm(y, sd[n + ".norm2.gamma"])))
        return y + h

    x = conv("post_quant_conv", z.float())
    x = conv("decoder.conv_in", x)
    x = res("decoder.mid_block.resnets.0", x)
    x = attention_block(sd, "decoder.mid_block.attentions.0", x)
    x = res("decoder.mid_block.resnets.1", x)
    stages["mid"] = x
    for i, (_, _, mode) in enumerate(cfg.up_plan()):
        for j in range(cfg.num_res_blocks + 1):
            x = res(f"decoder.up_blocks.{i}.resnets.{j}", x)
        n = f"decoder.up_blocks.{i}.upsamplers.0"
        if mode == "upsample3d" and x.shape[2] > 1:
            # frame 0 is kept; frames 1.. form their own causal sequence for time_conv (zero frames in front, NOT frame 0)
            tail = _interleave(conv(n + ".time_conv", x[:, :, 1:]))
            x = torch.cat([x[:, :, :1], tail], dim=2)
        if mode is not None:
            x = spatial_up_conv(sd, n, x)
        stages[f"up{i}"] = x
    x = conv("decoder.conv_out", F.silu(rms_norm(x, sd["decoder.norm_out.gamma"])))
    x = x.clamp(-1.0, 1.0)
    return (x, stages) if return_stages else x


# ------------------------------------------------------------------------------------------------ published chunked form
def decode_chunked(sd: Dict[str, torch.Tensor], cfg: WanVAEConfig, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKLWan._decode: one latent frame per decoder call, `feat_cache` of the last CACHE_T input frames of every causal conv."""
    cache: List[object] = [None] * 256

    def cconv(n, x, idx):
        """causal conv with the cache protocol of WanResidualBlock / WanDecoder3d"""
        i = idx[0]
        cx = x[:, :, -CACHE_T:].clone()
        if cx.shape[2] < 2 and cache[i] is not None:
            cx = torch.cat([cache[i][:, :, -1:].to(cx.device), cx], dim=2)
        y = causal_conv3d(x, sd[n + ".weight"], sd[n + ".bias"], cache[i])
        cache[i] = cx
        idx[0] += 1
        return y

    def res(n, x, idx):
        h = causal_conv3d(x, sd[n + ".conv_shortcut.weight"], sd[n + ".conv_shortcut.bias"]) if (n + ".conv_shortcut.weight") in sd else x
        y = cconv(n + ".conv1", F.silu(rms_norm(x, sd[n + ".norm1.gamma"])), idx)
        y = cconv(n + ".conv2", F.silu(rms_norm(y, sd[n + ".norm2.gamma"])), idx)
        return y + h

    def resample3d(n, x, idx):
        i = idx[0]
        if cache[i] is None:
            cache[i] = "Rep"
            idx[0] += 1
            return x
        cx = x[:, :, -CACHE_T:].clone()
        if cx.shape[2] < 2 and not isinstance(cache[i], str):
            cx = torch.cat([cache[i][:, :, -1:], cx], dim=2)
        if cx.shape[2] < 2 and isinstance(cache[i], str):
            cx = torch.cat([torch.zeros_like(cx), cx], dim=2)
        y = causal_conv3d(x, sd[n + ".time_conv.weight"], sd[n + ".time_conv.bias"], None if isinstance(cache[i], str) else cache[i])
        cache[i] = cx
        idx[0] += 1
        return _interleave(y)

    def decoder(x):
        idx = [0]
        x = cconv("decoder.conv_in", x, idx)
        x = res("decoder.mid_block.resnets.0", x, idx)
        x = attention_block(sd, "decoder.mid_block.attentions.0", x)
        x = res("decoder.mid_block.resnets.1", x, idx)
        for i, (_, _, mode) in enumerate(cfg.up_plan()):
            for j in range(cfg.num_res_blocks + 1):
                x = res(f"decoder.up_blocks.{i}.resnets.{j}", x, idx)
            n = f"decoder.up_blocks.{i}.upsamplers.0"
            if mode == "upsample3d":
                x = resample3d(n, x, idx)
            if mode is not None:
                x = spatial_up_conv(sd, n, x)
        return cconv("decoder.conv_out", F.silu(rms_norm(x, sd["decoder.norm_out.gamma"])), idx)

    x = causal_conv3d(z.float(), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    outs = [decoder(x[:, :, i:i + 1]) for i in range(x.shape[2])]
    return torch.cat(outs, dim=2).clamp(-1.0, 1.0)


# ------------------------------------------------------------------------------------------------ adapter-level control flow
def denormalise(latents: torch.Tensor, cfg: WanVAEConfig) -> torch.Tensor:
    """wan2_t2v.py:217-226: latents.float() / (1 / std) + mean."""
    mean = torch.tensor(cfg.latents_mean).view(1, cfg.z_dim, 1, 1, 1)
    inv = 1.0 / torch.tensor(cfg.latents_std).view(1, cfg.z_dim, 1, 1, 1)
    return latents.float() / inv + mean


def decode_latents(sd, cfg: WanVAEConfig, latents: torch.Tensor, postprocess: bool = True) -> torch.Tensor:
    """(B, 16, T, h, w) stored latents -> video_processor.postprocess_video(..., 'pt'): (B, F, 3, H, W) in [0, 1] (or the raw
    (B, 3, F, H, W) decoder output in [-1, 1])."""
    v = decode_full(sd, cfg, denormalise(latents, cfg))
    if not postprocess:
        return v
    return (v * 0.5 + 0.5).clamp(0.0, 1.0).permute(0, 2, 1, 3, 4)


def decode_flops(cfg: WanVAEConfig, T: int, h: int, w: int) -> float:
    """Algorithmic FLOPs (2 per MAC) of the convolutions and the mid attention of one decode of one sample.  Temporal taps that only
    ever meet the zero frames in front of a sequence are not counted (frame 0 sees 1 slice of a 3x3x3 kernel, frame 1 two, later ones
    three): a single latent frame costs a 2-D decode."""
    top = cfg.dims[0]
    slices = lambda f: 1 if f == 1 else 3 * f - 3                  # (temporal slice, frame) pairs with data under a causal kt = 3 kernel
    c3 = lambda f, hw, ci, co: slices(f) * hw * 9 * ci * co        # causal 3x3x3
    res = lambda f, hw, ci, co: c3(f, hw, ci, co) + c3(f, hw, co, co) + (f * hw * ci * co if ci != co else 0)
    fl = c3(T, h * w, cfg.z_dim, top)
    fl += 2 * res(T, h * w, top, top) + T * h * w * 4 * top * top + T * 2 * (h * w) ** 2 * top
    frames, hh, ww = T, h, w
    for ci, co, mode in cfg.up_plan():
        fl += res(frames, hh * ww, ci, co) + cfg.num_res_blocks * res(frames, hh * ww, co, co)
        if mode == "upsample3d" and frames > 1:
            fl += slices(frames - 1) * hh * ww * co * 2 * co
            frames = 2 * frames - 1
        if mode is not None:
            hh, ww = 2 * hh, 2 * ww
            fl += frames * hh * ww * 9 * co * (co // 2)
    fl += c3(frames, hh * ww, cfg.dims[-1], cfg.out_channels)
    return 2.0 * fl
