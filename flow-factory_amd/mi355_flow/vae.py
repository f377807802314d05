"""Native VAE decode (mi355_vae_*): host mirror of `SD3_5Adapter.decode_latents`
(reference src/flow_factory/models/stable_diffusion/sd3_5.py:161-172):

    latents = latents / vae.config.scaling_factor + vae.config.shift_factor
    images  = vae.decode(latents).sample ; images = image_processor.postprocess(images, output_type='pt')

The decoder body (diffusers AutoencoderKL) runs as NHWC implicit-GEMM convolutions in libmi355flow.so; there is no
PyTorch / CPU fallback.  Weights are bound from the HF state dict (`decoder.*` names; `post_quant_conv` is absent in SD3).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from . import _lib
from ._lib import VaeCfg
from .engine import WeightHolder, _ptr, _stream, dtype_code


@dataclass
class VAEConfig:
    """AutoencoderKL config fields the decoder needs (SD3 / SD3.5 defaults)."""
    latent_channels: int = 16
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    eps: float = 1e-6
    scaling_factor: float = 1.5305
    shift_factor: float = 0.0609

    @classmethod
    def from_hf(cls, config) -> "VAEConfig":
        g = (lambda k, d: getattr(config, k, d)) if not isinstance(config, dict) else (lambda k, d: config.get(k, d))
        if g("use_post_quant_conv", False):
            raise ValueError("mi355_flow: VAEs with post_quant_conv are not supported (SD3-family decoders have none)")
        return cls(g("latent_channels", 16), g("out_channels", 3), tuple(g("block_out_channels", (128, 256, 512, 512))),
                   g("layers_per_block", 2), g("norm_num_groups", 32), 1e-6, float(g("scaling_factor", 1.5305)),
                   float(g("shift_factor", 0.0609) or 0.0))

    def to_c(self) -> VaeCfg:
        if len(self.block_out_channels) > 8:
            raise ValueError("mi355_flow: at most 8 decoder blocks")
        boc = (C.c_int32 * 8)(*self.block_out_channels)
        return VaeCfg(self.latent_channels, self.out_channels, len(self.block_out_channels), self.layers_per_block,
                      self.norm_num_groups, boc, self.eps, self.scaling_factor, self.shift_factor)

    @property
    def spatial_scale(self) -> int:
        return 2 ** (len(self.block_out_channels) - 1)


class VAEDecoder(WeightHolder):
    """Owns the repacked bf16 decoder weights (mi355_vae) and per-shape workspaces (mi355_vae_plan)."""

    def __init__(self, cfg: VAEConfig = VAEConfig()):
        self.lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        c = cfg.to_c()
        _lib.check(self.lib.mi355_vae_create(C.byref(c), C.byref(h)), "vae_create")
        self._h = h
        self._plans: Dict[tuple, C.c_void_p] = {}

    _ABI, _WHAT = "vae", "VAE decoder"

    def _plan(self, batch: int, h: int, w: int) -> C.c_void_p:
        key = (h, w)
        ent = self._plans.get(key)
        if ent is None or ent[0] < batch:
            if ent is not None:
                self.lib.mi355_vae_plan_destroy(ent[1])
            p = C.c_void_p()
            _lib.check(self.lib.mi355_vae_plan_create(self._h, batch, h, w, C.byref(p)), "vae_plan_create")
            ent = (batch, p)
            self._plans[key] = ent
        return ent[1]

    def workspace_bytes(self, batch: int, h: int, w: int) -> int:
        return int(self.lib.mi355_vae_plan_workspace_bytes(self._plan(batch, h, w)))

    def decode(self, latents: torch.Tensor, postprocess: bool = True, out_dtype: torch.dtype = torch.bfloat16,
               max_batch: int = 4) -> torch.Tensor:
        """latents (B, C, h, w) in any of fp32 / bf16 / fp16 -> images (B, 3, 8h, 8w) in [0, 1] (postprocess) .
        Decodes `max_batch` images per launch sequence to bound the workspace (~2 GiB per 1024^2 image)."""
        if latents.dim() != 4 or latents.shape[1] != self.cfg.latent_channels:
            raise ValueError(f"mi355_flow: expected latents (B, {self.cfg.latent_channels}, h, w), got {tuple(latents.shape)}")
        if out_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("mi355_flow: images are produced in float32 or bfloat16")
        B, _, h, w = latents.shape
        s = self.cfg.spatial_scale
        latents = latents.contiguous()
        out = torch.empty((B, self.cfg.out_channels, h * s, w * s), dtype=out_dtype, device=latents.device)
        mb = max(1, min(max_batch, B))
        plan = self._plan(mb, h, w)
        st = _stream()
        for b0 in range(0, B, mb):
            n = min(mb, B - b0)
            _lib.check(self.lib.mi355_vae_decode(plan, st, _ptr(latents[b0:b0 + n]), dtype_code(latents.dtype), n,
                                                 _ptr(out[b0:b0 + n]), dtype_code(out_dtype), int(postprocess)), "vae_decode")
        return out

    def close(self) -> None:
        for _, p in self._plans.values():
            self.lib.mi355_vae_plan_destroy(p)
        self._plans.clear()
        if self._h:
            self.lib.mi355_vae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------- operator-level wrappers (tests, microbench)
def op_conv_repack(weight: torch.Tensor, cin_pad: int = 0) -> torch.Tensor:
    """torch conv / linear weight [Co][Ci][kh][kw] (or [Co][Ci]) -> packed bf16 [Co][taps][Ci_pad]."""
    lib = _lib.load()
    weight = weight.contiguous()
    co, ci = weight.shape[0], weight.shape[1]
    taps = weight.numel() // (co * ci)
    cpad = cin_pad or (ci + 63) // 64 * 64
    out = torch.empty((co, taps, cpad), dtype=torch.bfloat16, device=weight.device)
    _lib.check(lib.mi355_op_conv_repack(_stream(), _ptr(weight), dtype_code(weight.dtype), _ptr(out), co, ci, cpad, taps), "op_conv_repack")
    return out


def op_conv3x3(x_nhwc: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, residual: torch.Tensor = None,
               upsample: bool = False) -> torch.Tensor:
    """x (B, Hin, Win, Cin) bf16 NHWC -> (B, H, W, Co) bf16, padding 1; `upsample` folds a nearest-2x resize of x in."""
    lib = _lib.load()
    B, Hin, Win, Cin = x_nhwc.shape
    H, W = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    Co = w_packed.shape[0]
    out = torch.empty((B, H, W, Co), dtype=torch.bfloat16, device=x_nhwc.device)
    _lib.check(lib.mi355_op_conv3x3(_stream(), _ptr(x_nhwc), _ptr(w_packed), _ptr(bias), _ptr(residual), _ptr(out), B, H, W, Cin, Co,
                                    int(upsample)), "op_conv3x3")
    return out


def op_group_norm(x_nhwc: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float = 1e-6,
                  silu: bool = False) -> torch.Tensor:
    lib = _lib.load()
    B, C = x_nhwc.shape[0], x_nhwc.shape[-1]
    HW = x_nhwc.numel() // (B * C)
    out = torch.empty_like(x_nhwc)
    scratch = torch.empty(B * (2048 * C + 2 * C), dtype=torch.float32, device=x_nhwc.device)
    _lib.check(lib.mi355_op_group_norm(_stream(), _ptr(x_nhwc), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(scratch), B, HW, C, groups,
                                       eps, int(silu)), "op_group_norm")
    return out


# ----------------------------------------------------------------------------- causal 3-D video VAE (Wan / Qwen-Image)
@dataclass
class WanVAEConfig:
    """diffusers AutoencoderKLWan / AutoencoderKLQwenImage config fields the decoder needs (Wan2.1 defaults)."""
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

    @classmethod
    def from_hf(cls, config) -> "WanVAEConfig":
        g = (lambda k, d: getattr(config, k, d)) if not isinstance(config, dict) else (lambda k, d: config.get(k, d))
        if g("is_residual", False) or int(g("patch_size", 1) or 1) != 1:
            raise ValueError("mi355_flow: the residual / patchified Wan2.2-TI2V VAE (is_residual, patch_size) is not supported")
        if tuple(g("attn_scales", ()) or ()):
            raise ValueError("mi355_flow: attn_scales is not supported (the released Wan / Qwen-Image VAEs have none)")
        return cls(int(g("z_dim", 16)), int(g("base_dim", 96)), tuple(g("dim_mult", (1, 2, 4, 4))), int(g("num_res_blocks", 2)),
                   tuple(bool(v) for v in g("temperal_downsample", (False, True, True))), int(g("out_channels", 3) or 3),
                   tuple(float(v) for v in g("latents_mean", cls.latents_mean)), tuple(float(v) for v in g("latents_std", cls.latents_std)))

    def to_c(self):
        from ._lib import WvaeCfg
        if len(self.dim_mult) != 4 or len(self.temperal_downsample) != 3:
            raise ValueError("mi355_flow: the video VAE engine expects 4 stages (dim_mult) and 3 temporal flags")
        if len(self.latents_mean) != self.z_dim or len(self.latents_std) != self.z_dim or self.z_dim > 16:
            raise ValueError("mi355_flow: latents_mean / latents_std must have z_dim <= 16 entries")
        pad = lambda v: tuple(v) + (0.0,) * (16 - len(v))
        return WvaeCfg(self.z_dim, self.base_dim, self.num_res_blocks, self.out_channels, (C.c_int32 * 4)(*self.dim_mult),
                       (C.c_int32 * 3)(*[int(b) for b in self.temperal_downsample[::-1]]), (C.c_float * 16)(*pad(self.latents_mean)),
                       (C.c_float * 16)(*pad(self.latents_std)))

    def num_frames(self, latent_t: int) -> int:
        f = latent_t
        for up in self.temperal_downsample[::-1]:
            if up and f > 1:
                f = 2 * f - 1
        return f


class WanVAEDecoder(WeightHolder):
    """Owns the repacked bf16 weights of the causal 3-D decoder (mi355_wvae) and per-shape workspaces.  Bind the HF state dict of the
    VAE (`post_quant_conv.*`, `decoder.*`; encoder keys are ignored with `partial`-free strictness on the decoder names)."""

    _ABI, _WHAT = "wvae", "video VAE decoder"

    def __init__(self, cfg: WanVAEConfig = WanVAEConfig()):
        self.lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        c = cfg.to_c()
        _lib.check(self.lib.mi355_wvae_create(C.byref(c), C.byref(h)), "wvae_create")
        self._h = h
        self._plans: Dict[tuple, tuple] = {}

    def _plan(self, batch: int, t: int, h: int, w: int) -> C.c_void_p:
        key = (t, h, w)
        ent = self._plans.get(key)
        if ent is None or ent[0] < batch:
            if ent is not None:
                self.lib.mi355_wvae_plan_destroy(ent[1])
            p = C.c_void_p()
            _lib.check(self.lib.mi355_wvae_plan_create(self._h, batch, t, h, w, C.byref(p)), "wvae_plan_create")
            ent = (batch, p)
            self._plans[key] = ent
        return ent[1]

    def workspace_bytes(self, batch: int, t: int, h: int, w: int) -> int:
        return int(self.lib.mi355_wvae_plan_workspace_bytes(self._plan(batch, t, h, w)))

    def decode(self, latents: torch.Tensor, postprocess: bool = True, out_dtype: torch.dtype = torch.bfloat16, max_batch: int = 1,
               denormalise: bool = True) -> torch.Tensor:
        """latents (B, 16, T, h, w) in fp32 / bf16 / fp16 -> video (B, F, 3, 8h, 8w), F = 1 + 4 (T - 1): in [0, 1] with `postprocess`
        (= `VideoProcessor.postprocess_video(vae.decode(z), 'pt')`), else the decoder output in [-1, 1] (permute(0, 2, 1, 3, 4) of
        `vae.decode(z)`).  `denormalise`: apply `z = latents / (1 / std) + mean` first, as the adapters do (wan2_t2v.py:217-226)."""
        if latents.dim() != 5 or latents.shape[1] != self.cfg.z_dim:
            raise ValueError(f"mi355_flow: expected latents (B, {self.cfg.z_dim}, T, h, w), got {tuple(latents.shape)}")
        if out_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("mi355_flow: videos are produced in float32 or bfloat16")
        B, _, T, h, w = latents.shape
        F = self.cfg.num_frames(T)
        latents = latents.contiguous()
        out = torch.empty((B, F, self.cfg.out_channels, h * 8, w * 8), dtype=out_dtype, device=latents.device)
        mb = max(1, min(max_batch, B))
        plan = self._plan(mb, T, h, w)
        st = _stream()
        for b0 in range(0, B, mb):
            n = min(mb, B - b0)
            _lib.check(self.lib.mi355_wvae_decode(plan, st, _ptr(latents[b0:b0 + n]), dtype_code(latents.dtype), n, _ptr(out[b0:b0 + n]),
                                                  dtype_code(out_dtype), int(postprocess), int(denormalise)), "wvae_decode")
        return out

    def close(self) -> None:
        for _, p in self._plans.values():
            self.lib.mi355_wvae_plan_destroy(p)
        self._plans.clear()
        if self._h:
            self.lib.mi355_wvae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def op_conv3d_causal(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, kt: int, ks: int, frames: int = None, residual: torch.Tensor = None,
                     upsample: bool = False, skip_frames: int = 0) -> torch.Tensor:
    """x (B, T_in, Hin, Win, Cin) bf16 -> (B, T, H, W, Co) bf16: causal conv with kt temporal / ks x ks spatial taps over the frames
    [skip_frames, skip_frames + T) of each sample taken as a sequence of their own (zeros before its first frame)."""
    lib = _lib.load()
    B, T_in, Hin, Win, Cin = x.shape
    T = frames if frames is not None else T_in - skip_frames
    H, W = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    Co = w_packed.shape[0]
    out = torch.empty((B, T, H, W, Co), dtype=torch.bfloat16, device=x.device)
    src = x.data_ptr() + skip_frames * Hin * Win * Cin * 2
    _lib.check(lib.mi355_op_conv3d_causal(_stream(), src, _ptr(w_packed), _ptr(bias), _ptr(residual), _ptr(out), B, T, T_in, H, W, Cin, Co,
                                          kt, ks, int(upsample)), "op_conv3d_causal")
    return out


def op_wan_rms(x: torch.Tensor, gamma: torch.Tensor, channels: int, silu: bool = False) -> torch.Tensor:
    """x (..., C_pad) bf16, gamma fp32 (C_pad,) zero beyond `channels`."""
    lib = _lib.load()
    out = torch.empty_like(x)
    _lib.check(lib.mi355_op_wan_rms(_stream(), _ptr(x), _ptr(gamma), _ptr(out), x.numel() // x.shape[-1], channels, x.shape[-1], int(silu)),
               "op_wan_rms")
    return out
