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
from typing import Dict, List, Tuple

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
