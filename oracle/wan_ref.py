"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU fp32 restatement of the Wan2.1 text-to-video transformer forward and of the Flow-Factory Wan T2V rollout step
(reference src/flow_factory/models/wan/wan2_t2v.py:234-421 `inference`, :426-543 `forward`:
latents (B, 16, T, h, w) fp32 -> storage dtype, `timestep = t.expand(B)` with t the scheduler's INTEGER timestep, separate
cond / uncond passes combined as `u + g (c - u)`, then `UniPCMultistepSDEScheduler.step`, whose train / rollout branch
(scheduler/unipc_multistep.py:296-421) is the same four dynamics as the flow-match Euler SDE step with sigma = t / 1000).

The rollout CONTROL FLOW (`rollout`, `rollout_two_expert`) is PINNED bit for bit against the reference's own `Wan2_T2V_Adapter` and
`UniPCMultistepSDEScheduler` (tests/test_rollout_control_flow_pin.py, oracle/make_rollout_golden.py).
NETWORK BODY: PARITY UNPINNED: the model body is diffusers' `WanTransformer3DModel` (un-vendored third-party dependency); restated from the
published architecture with HF state-dict names:
  patch_embedding Conv3d(16, D, k = s = (1, 2, 2)); WanRotaryPosEmbed (head_dim 128 split t/h/w = 44/42/42, adjacent pairs, float64
  angles); condition_embedder {time_embedder (sinusoidal 256 -> D -> D), time_proj Linear(D, 6D) on silu(temb), text_embedder
  (4096 -> D -> D, gelu-tanh)}; N x WanTransformerBlock: modulation = scale_shift_table[1, 6, D] + time_proj (fp32),
  self-attention (q/k RMSNorm ACROSS heads, RoPE) gated, cross-attention to the text tokens (affine LayerNorm before, q/k RMSNorm
  across heads, no RoPE, no gate), gelu-tanh feed-forward gated; output norm modulated by scale_shift_table[1, 2, D] + temb, proj_out,
  un-patchify with feature order (p_t, p_h, p_w, c).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class WanConfig:
    in_channels: int = 16
    out_channels: int = 16
    num_layers: int = 30
    num_attention_heads: int = 12
    attention_head_dim: int = 128
    ffn_dim: int = 8960
    text_dim: int = 4096
    freq_dim: int = 256
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


WAN21_T2V_1_3B = WanConfig()


def tiny_config(num_layers=2, heads=2, ffn_dim=512, text_dim=128) -> WanConfig:
    return WanConfig(num_layers=num_layers, num_attention_heads=heads, ffn_dim=ffn_dim, text_dim=text_dim)


def state_dict_shapes(cfg: WanConfig) -> Dict[str, Tuple[int, ...]]:
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


def make_synthetic_state_dict(cfg: WanConfig, seed: int = 55, std: float = 0.03) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, s in state_dict_shapes(cfg).items():
        t = torch.randn(s, generator=g) * std
        if ".norm_q." in n or ".norm_k." in n or n.endswith("norm2.weight"):
            t = t + 1.0
        if "scale_shift_table" in n:
            t = t * (0.3 / std)          # learned tables are O(1/sqrt(D)) .. O(1); make the modulation matter
        sd[n] = t
    return sd


def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([a.cos(), a.sin()], dim=-1)


def rope_axes(head_dim: int = 128) -> Tuple[int, int, int]:
    """WanRotaryPosEmbed: h_dim = w_dim = 2 * (head_dim // 6), t_dim = head_dim - h_dim - w_dim."""
    hw = 2 * (head_dim // 6)
    return head_dim - 2 * hw, hw, hw


def rope_cos_sin(T: int, Hp: int, Wp: int, head_dim: int = 128, theta: float = 10000.0):
    """cos, sin (T*Hp*Wp, head_dim/2): pair j of a token uses its t / h / w index by axis; angles in float64."""
    dt, dh, dw = rope_axes(head_dim)
    parts = []
    for d, n in ((dt, T), (dh, Hp), (dw, Wp)):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        parts.append(torch.outer(torch.arange(n, dtype=torch.float64), freqs))      # (n, d/2)
    at = parts[0][:, None, None, :].expand(T, Hp, Wp, -1)
    ah = parts[1][None, :, None, :].expand(T, Hp, Wp, -1)
    aw = parts[2][None, None, :, :].expand(T, Hp, Wp, -1)
    ang = torch.cat([at, ah, aw], dim=-1).reshape(T * Hp * Wp, head_dim // 2)
    return ang.cos().float(), ang.sin().float()


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x (B, H, S, D) as complex adjacent pairs times exp(i angle)."""
    xr = x.reshape(*x.shape[:-1], -1, 2)
    a, b = xr[..., 0], xr[..., 1]
    c, s = cos[None, None], sin[None, None]
    return torch.stack([a * c - b * s, b * c + a * s], dim=-1).flatten(3)


def patchify(lat: torch.Tensor) -> torch.Tensor:
    """(B, C, T, h, w) -> (B, T*h/2*w/2, C*4): Conv3d k = s = (1, 2, 2) as a matmul, feature = c*4 + ph*2 + pw."""
    B, C, T, h, w = lat.shape
    return lat.view(B, C, T, h // 2, 2, w // 2, 2).permute(0, 2, 3, 5, 1, 4, 6).reshape(B, T * (h // 2) * (w // 2), C * 4)


def unpatchify(x: torch.Tensor, T: int, h: int, w: int, C: int) -> torch.Tensor:
    """(B, S, 4*C) with feature order (p_h, p_w, c) -> (B, C, T, h, w)   (p_t = 1)."""
    B = x.shape[0]
    return x.view(B, T, h // 2, w // 2, 2, 2, C).permute(0, 6, 1, 2, 4, 3, 5).reshape(B, C, T, h, w)


def _id(x):
    return x


def _rms(x, w, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def _heads(x, H):
    B, S, D = x.shape
    return x.view(B, S, H, D // H).transpose(1, 2)


def wan_forward(sd: Dict[str, torch.Tensor], cfg: WanConfig, latents: torch.Tensor, timestep: torch.Tensor, enc: torch.Tensor,
                quant: Optional[Callable] = None) -> torch.Tensor:
    """latents (B, 16, T, h, w); timestep (B,) in [0, 1000]; enc (B, Nt, text_dim) -> velocity (B, 16, T, h, w).
    `timestep` of shape (B, S) = ONE TIMESTEP PER TOKEN (`expand_timesteps`, Wan2.2-TI2V-5B: the adapter passes `mask[0][0][:, ::2, ::2] * t`
    flattened, reference wan2_t2v.py:498-504): the time embedding and its six modulation vectors are then computed per token and applied per
    token (published WanTransformer3DModel: `timestep.ndim == 2` -> temb (B, S, D), timestep_proj (B, S, 6, D); model body not in the reference
    tree -- restated, unpinned).  With an all-ones mask every token carries the same t and the result equals the scalar-timestep forward,
    which is what the engine runs for TI2V in text-to-video use (tests/test_host_mirrors.py::test_wan_oracle_per_token_timesteps_*)."""
    q = quant or _id
    per_token = timestep.ndim == 2
    D, H, eps = cfg.dim, cfg.num_attention_heads, cfg.eps
    lin = lambda n, x: F.linear(q(x), q(sd[n + ".weight"]), sd[n + ".bias"])
    ln = lambda x: F.layer_norm(x, (D,), eps=eps)
    B, C, T, h, w = latents.shape
    cos, sin = rope_cos_sin(T, h // 2, w // 2, cfg.attention_head_dim)
    x = q(F.linear(q(patchify(latents.float())), q(sd["patch_embedding.weight"].reshape(D, -1)), sd["patch_embedding.bias"]))
    temb = q(lin("condition_embedder.time_embedder.linear_2",
                 F.silu(q(lin("condition_embedder.time_embedder.linear_1", q(timestep_embedding(timestep.float().flatten(), cfg.freq_dim)))))))
    tproj = q(lin("condition_embedder.time_proj", q(F.silu(temb))))
    if per_token:                                       # (B * S, .) -> one set of modulation vectors per token
        temb, tproj = temb.view(B, -1, D), tproj.view(B, -1, 6, D)
    else:
        tproj = tproj.view(B, 6, D)
    bc = (lambda v: v) if per_token else (lambda v: v[:, None])       # a (B, S, D) vector applies as is, a (B, D) one broadcasts over the tokens
    ctx = q(lin("condition_embedder.text_embedder.linear_2",
                q(F.gelu(lin("condition_embedder.text_embedder.linear_1", enc.float()), approximate="tanh"))))

    def attention(qq, kk, vv):
        o = F.scaled_dot_product_attention(q(qq), q(kk), q(vv))
        return q(o.transpose(1, 2).reshape(o.shape[0], o.shape[2], D))

    for i in range(cfg.num_layers):
        b = f"blocks.{i}"
        m = (sd[f"{b}.scale_shift_table"] + tproj.float()).unbind(-2)     # shift, scale, gate, c_shift, c_scale, c_gate  (B, D) / (B, S, D)
        xn = q(ln(x) * (1 + bc(m[1])) + bc(m[0]))
        qq = apply_rope(_heads(_rms(q(lin(f"{b}.attn1.to_q", xn)), sd[f"{b}.attn1.norm_q.weight"], eps), H), cos, sin)
        kk = apply_rope(_heads(_rms(q(lin(f"{b}.attn1.to_k", xn)), sd[f"{b}.attn1.norm_k.weight"], eps), H), cos, sin)
        vv = _heads(lin(f"{b}.attn1.to_v", xn), H)
        x = q(x + q(lin(f"{b}.attn1.to_out.0", attention(qq, kk, vv))) * bc(m[2]))
        xn = q(F.layer_norm(x, (D,), sd[f"{b}.norm2.weight"], sd[f"{b}.norm2.bias"], eps))
        qq = _heads(_rms(q(lin(f"{b}.attn2.to_q", xn)), sd[f"{b}.attn2.norm_q.weight"], eps), H)
        kk = _heads(_rms(q(lin(f"{b}.attn2.to_k", ctx)), sd[f"{b}.attn2.norm_k.weight"], eps), H)
        vv = _heads(lin(f"{b}.attn2.to_v", ctx), H)
        x = q(x + q(lin(f"{b}.attn2.to_out.0", attention(qq, kk, vv))))
        xn = q(ln(x) * (1 + bc(m[4])) + bc(m[3]))
        ff = q(lin(f"{b}.ffn.net.2", q(F.gelu(lin(f"{b}.ffn.net.0.proj", xn), approximate="tanh"))))
        x = q(x + ff * bc(m[5]))
    mo = (sd["scale_shift_table"] + temb.float().unsqueeze(-2)).unbind(-2)      # shift, scale
    xo = q(ln(x) * (1 + bc(mo[1])) + bc(mo[0]))
    out = q(lin("proj_out", xo))
    return unpatchify(out, T, h, w, cfg.out_channels)


def forward_flops(cfg: WanConfig, S: int, Nt: int) -> float:
    """Algorithmic matmul FLOPs of one forward for one sample (2 FLOP/MAC; conditioning MLPs excluded)."""
    D, Fd = cfg.dim, cfg.ffn_dim
    per_layer = S * (4 * D * D) + 2 * S * S * D                         # self-attention projections + QK^T + PV
    per_layer += S * 2 * D * D + Nt * 2 * D * D + 2 * S * Nt * D        # cross-attention: q/out on S rows, k/v on Nt rows, scores + PV
    per_layer += S * 2 * D * Fd
    emb = S * cfg.in_channels * 4 * D + S * D * cfg.out_channels * 4 + Nt * cfg.text_dim * D + Nt * D * D
    return 2.0 * (cfg.num_layers * per_layer + emb)


# ------------------------------------------------------------------ rollout control flow (wan2_t2v.py:344-376, :426-543)
def unipc_flow_schedule(num_inference_steps: int, flow_shift: float = 3.0, num_train_timesteps: int = 1000):
    """diffusers UniPCMultistepScheduler.set_timesteps(use_flow_sigmas=True), restated from the published algorithm:
    sigmas = flip(shift*s / (1 + (shift-1)*s)), s = 1 - linspace(1, 1/N_train, steps+1), last dropped; timesteps = int64(sigma*N_train)."""
    import numpy as np
    alphas = np.linspace(1, 1 / num_train_timesteps, num_inference_steps + 1)
    sig = 1.0 - alphas
    sig = np.flip(flow_shift * sig / (1 + (flow_shift - 1) * sig))[:-1].copy()
    timesteps = (sig * num_train_timesteps).copy().astype(np.int64)
    sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
    return torch.from_numpy(timesteps), torch.from_numpy(sigmas)


def rollout(sd, cfg: WanConfig, prompt_embeds, negative_prompt_embeds, guidance_scale, init_latents, step_noise, timesteps, sigmas,
            noise_levels, storage_dtype=torch.float16, dynamics_type="Flow-SDE", compute_log_prob=True, quant=None, denoiser=None):
    """N-step loop: latents (B,16,T,h,w); timesteps int64 (N,); sigma of a step = t / 1000 (unipc_multistep.py:288-291).
    `denoiser(hidden_states, timestep, encoder_hidden_states)` (tests/test_rollout_control_flow_pin.py) replaces the network AT THE
    ADAPTER'S CALL (wan2_t2v.py:505-523): latents in the transformer dtype, the integer timestep expanded to the batch."""
    from . import scheduler_ref as S
    from .rollout_ref import cfg_combine_bf16
    N = len(timesteps)
    lat = S.cast_latents(init_latents, storage_dtype)
    all_lat, lps, means = [lat], [], []
    sigma_max = float(sigmas[1])
    do_cfg = negative_prompt_embeds is not None and guidance_scale > 1.0
    B = lat.shape[0]
    for i in range(N):
        t = timesteps[i].float()
        t_next = timesteps[i + 1].float() if i + 1 < N else torch.tensor(0.0)
        eta = float(noise_levels[i])
        clp = compute_log_prob and eta > 0
        if denoiser is not None:
            net = lambda emb: denoiser(lat.to(torch.bfloat16), timesteps[i].expand(B), emb).to(torch.bfloat16)      # noqa: E731
        else:
            x_in = lat.to(torch.bfloat16).float()
            net = lambda emb: wan_forward(sd, cfg, x_in, t.expand(B), emb.float(), quant=quant).to(torch.bfloat16)    # noqa: E731
        v = net(prompt_embeds)
        if do_cfg:
            v = cfg_combine_bf16(net(negative_prompt_embeds), v, guidance_scale)
        out = S.sde_step(v, lat, t / 1000, t_next / 1000, eta, dynamics_type=dynamics_type, sigma_max=sigma_max,
                         variance_noise=step_noise[i], compute_log_prob=clp)
        lat = S.cast_latents(out["next_latents"], storage_dtype)
        all_lat.append(lat)
        lps.append(out["log_prob"] if clp else torch.full((B,), float("nan")))
        means.append(out["next_latents_mean"])
    return dict(all_latents=torch.stack(all_lat, 0), log_probs=torch.stack(lps, 0), next_latents_means=torch.stack(means, 0))


def rollout_two_expert(sd_hi, sd_lo, cfg: WanConfig, boundary_timestep, prompt_embeds, negative_prompt_embeds, guidance_scale, guidance_scale_2,
                       init_latents, step_noise, timesteps, sigmas, noise_levels, storage_dtype=torch.float16, dynamics_type="Flow-SDE",
                       compute_log_prob=True, denoisers=None):
    """Wan2.2 two-expert loop (reference wan2_t2v.py:476-487 inside the loop of :344-376): the high-noise expert with `guidance_scale`
    while t >= boundary_timestep, the low-noise expert with `guidance_scale_2` below; CFG is decided per expert."""
    from . import scheduler_ref as S
    from .rollout_ref import cfg_combine_bf16
    N = len(timesteps)
    lat = S.cast_latents(init_latents, storage_dtype)
    all_lat, lps, means = [lat], [], []
    sigma_max = float(sigmas[1])
    B = lat.shape[0]
    for i in range(N):
        t = timesteps[i].float()
        t_next = timesteps[i + 1].float() if i + 1 < N else torch.tensor(0.0)
        hi = float(t) >= boundary_timestep
        sd, g = (sd_hi, guidance_scale) if hi else (sd_lo, guidance_scale_2)
        eta = float(noise_levels[i])
        clp = compute_log_prob and eta > 0
        if denoisers is not None:             # (high-noise, low-noise) stand-ins at the adapter's transformer call
            net = lambda emb: denoisers[0 if hi else 1](lat.to(torch.bfloat16), timesteps[i].expand(B), emb).to(torch.bfloat16)    # noqa: E731
        else:
            x_in = lat.to(torch.bfloat16).float()
            net = lambda emb: wan_forward(sd, cfg, x_in, t.expand(B), emb.float()).to(torch.bfloat16)                            # noqa: E731
        v = net(prompt_embeds)
        if negative_prompt_embeds is not None and g > 1.0:
            v = cfg_combine_bf16(net(negative_prompt_embeds), v, g)
        out = S.sde_step(v, lat, t / 1000, t_next / 1000, eta, dynamics_type=dynamics_type, sigma_max=sigma_max,
                         variance_noise=step_noise[i], compute_log_prob=clp)
        lat = S.cast_latents(out["next_latents"], storage_dtype)
        all_lat.append(lat)
        lps.append(out["log_prob"] if clp else torch.full((B,), float("nan")))
        means.append(out["next_latents_mean"])
    return dict(all_latents=torch.stack(all_lat, 0), log_probs=torch.stack(lps, 0), next_latents_means=torch.stack(means, 0))


def rollout_eval(sd, cfg: WanConfig, prompt_embeds, negative_prompt_embeds, guidance_scale, init_latents, timesteps, sigmas,
                 storage_dtype=torch.float16, solver_order: int = 2, solver_type: str = "bh2", sd_low=None, boundary_timestep=None,
                 guidance_scale_2=None):
    """EVALUATION-mode sampling: the adapter loop of reference wan2_t2v.py:346-375 with `scheduler.step` in its `is_eval` branch
    (scheduler/unipc_multistep.py:282-285 -> diffusers' UniPCMultistepScheduler.step, restated in oracle/unipc_ref.py: PARITY UNPINNED).
    No noise is drawn; the network sees the latents in bf16 and the integer timestep; the cond / uncond passes are combined in bf16; the
    solver receives the bf16 prediction and the latents in their storage dtype.  `sd_low` / `boundary_timestep` / `guidance_scale_2`: the
    Wan2.2 two-expert rule (wan2_t2v.py:476-487)."""
    from . import scheduler_ref as S
    from .rollout_ref import cfg_combine_bf16
    from .unipc_ref import UniPCRef
    N = len(timesteps)
    solver = UniPCRef([float(s) for s in sigmas], solver_order=solver_order, solver_type=solver_type)
    lat = S.cast_latents(init_latents, storage_dtype)
    all_lat = [lat]
    B = lat.shape[0]
    for i in range(N):
        t = timesteps[i].float()
        low = sd_low is not None and boundary_timestep is not None and float(t) < boundary_timestep
        sd_i, g = (sd_low, guidance_scale_2 if guidance_scale_2 is not None else guidance_scale) if low else (sd, guidance_scale)
        x_in = lat.to(torch.bfloat16).float()
        net = lambda emb: wan_forward(sd_i, cfg, x_in, t.expand(B), emb.float()).to(torch.bfloat16)          # noqa: E731
        v = net(prompt_embeds)
        if negative_prompt_embeds is not None and g > 1.0:
            v = cfg_combine_bf16(net(negative_prompt_embeds), v, g)
        lat = S.cast_latents(solver.step(v, lat), storage_dtype)
        all_lat.append(lat)
    return dict(all_latents=torch.stack(all_lat, 0))
