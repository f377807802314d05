"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (torch fp32 / numpy) of the scheduler half of the hot path:

  * schedule        -- reference src/flow_factory/scheduler/flow_match_euler_discrete.py:37-77
                       (+ diffusers FlowMatchEulerDiscreteScheduler.set_timesteps, restated in
                       oracle/diffusers_stub.py)
  * SDE-step select -- same file :126-198
  * step()          -- same file :243-438  (ODE / Flow-SDE / Dance-SDE / CPS + Gaussian log-prob)
  * cast_latents    -- reference src/flow_factory/models/abc.py:172-182

Pinned: `tests/test_oracle_golden.py` checks every function here against fixtures that
`oracle/make_golden.py` produced by executing the reference's OWN code
(tests/golden/scheduler_*.npz).  Unlike the reference this file has no host syncs
and works on explicit (sigma, sigma_next, noise_level) scalars, which is also what the
HIP kernel consumes.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

DYNAMICS = ("ODE", "Flow-SDE", "Dance-SDE", "CPS")


# --------------------------------------------------------------------------- schedule
def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    """flow_match_euler_discrete.py:37-47."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def make_schedule(
    num_inference_steps: int,
    shift: float = 3.0,
    use_dynamic_shifting: bool = False,
    seq_len: Optional[int] = None,
    num_train_timesteps: int = 1000,
):
    """flow_match_euler_discrete.py:49-77 -> diffusers set_timesteps(sigmas=linspace(1,1/N,N), mu).

    Returns (timesteps[N] f32, sigmas[N+1] f32) as torch tensors; arithmetic in float32 numpy
    exactly like diffusers (np.float32 array, python-float scalars)."""
    sig = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps).astype(np.float32)
    if use_dynamic_shifting:
        mu = calculate_shift(seq_len)
        sig = math.exp(mu) / (math.exp(mu) + (1 / sig - 1) ** 1.0)
    else:
        sig = shift * sig / (1 + (shift - 1) * sig)
    sig_t = torch.from_numpy(np.asarray(sig)).to(torch.float32)
    timesteps = sig_t * num_train_timesteps
    sigmas = torch.cat([sig_t, torch.zeros(1)])
    return timesteps, sigmas


def current_sde_steps(sde_steps: Optional[Sequence[int]], num_sde_steps: Optional[int], seed: int, num_timesteps: int):
    """flow_match_euler_discrete.py:126-165: seeded choice of `num_sde_steps` of `sde_steps`."""
    steps = (
        torch.tensor(list(sde_steps), dtype=torch.int64)
        if sde_steps is not None
        else torch.arange(0, num_timesteps - 1, dtype=torch.int64)
    )
    n = num_sde_steps if num_sde_steps is not None else len(steps)
    if n >= len(steps):
        return steps
    g = torch.Generator().manual_seed(seed)
    sel = torch.randperm(len(steps), generator=g)[:n]
    return steps[sel]


def noise_levels(num_timesteps: int, sde_idx: torch.Tensor, noise_level: float) -> torch.Tensor:
    """flow_match_euler_discrete.py:181-185."""
    out = torch.zeros(num_timesteps, dtype=torch.float32)
    out[sde_idx] = noise_level
    return out


# --------------------------------------------------------------------------- cast
def cast_latents(latents: torch.Tensor, target: Optional[torch.dtype]) -> torch.Tensor:
    """models/abc.py:172-182 (fp16 clamp at +-65504, then cast)."""
    if target is None or latents.dtype == target:
        return latents
    if target == torch.float16:
        latents = latents.clamp(-65504.0, 65504.0)
    return latents.to(target)


# --------------------------------------------------------------------------- step
def _bcast(v, ref: torch.Tensor) -> torch.Tensor:
    """utils/base.py:358-376 `to_broadcast_tensor`."""
    if not isinstance(v, torch.Tensor):
        v = torch.tensor(v if isinstance(v, list) else [v])
    v = v.to(device=ref.device, dtype=ref.dtype)
    if v.numel() == 1:
        v = v.expand(ref.shape[0])
    return v.view(-1, *([1] * (ref.dim() - 1)))


def sde_step(
    noise_pred: torch.Tensor,
    latents: torch.Tensor,
    sigma,
    sigma_next,
    noise_level,
    dynamics_type: str = "Flow-SDE",
    sigma_max: float = None,
    variance_noise: Optional[torch.Tensor] = None,
    next_latents: Optional[torch.Tensor] = None,
    compute_log_prob: bool = True,
):
    """flow_match_euler_discrete.py:305-426.

    `sigma = t/1000`, `sigma_next = t_next/1000` (:302-303).  `variance_noise` replaces the
    `randn_tensor` draw (:352-357) so that the caller controls RNG order.  Returns a dict with
    next_latents (fp32, value-rounded to the input dtype when freshly sampled, :362),
    next_latents_mean, std_dev_t, dt, log_prob (or None)."""
    assert dynamics_type in DYNAMICS
    in_dtype = latents.dtype
    v = noise_pred.float()
    x = latents.float()
    nxt = next_latents.float() if next_latents is not None else None
    if dynamics_type == "ODE":
        noise_level = 0.0
    eta = _bcast(noise_level, x)
    s = _bcast(sigma, x)
    sp = _bcast(sigma_next, x)
    dt = sp - s
    log_prob = None
    if dynamics_type == "ODE":  # :329-340
        mean = x + v * dt
        std = torch.zeros_like(s)
        if nxt is None:
            nxt = mean
        if compute_log_prob:
            log_prob = torch.zeros(x.shape[0], dtype=torch.float32)
    elif dynamics_type == "Flow-SDE":  # :342-371
        smax = _bcast(sigma_max, x)
        std = torch.sqrt(s / (1 - torch.where(s == 1.0, smax, s))) * eta
        mean = x * (1 + std**2 / (2 * s) * dt) + v * (1 + std**2 * (1 - s) / (2 * s)) * dt
        if nxt is None:
            nxt = mean + std * torch.sqrt(-1 * dt) * variance_noise.float()
            nxt = nxt.to(in_dtype).float()
        if compute_log_prob:
            sv = std * torch.sqrt(-1 * dt)
            lp = -((nxt - mean) ** 2) / (2 * sv**2) - torch.log(sv) - torch.log(torch.sqrt(2 * torch.as_tensor(math.pi)))
            log_prob = lp.mean(dim=tuple(range(1, lp.ndim)))
    elif dynamics_type == "Dance-SDE":  # :373-398
        x0 = x - s * v
        std = eta
        log_term = 0.5 * eta**2 * (x - x0 * (1 - s)) / s**2
        mean = x + (v + log_term) * dt
        if nxt is None:
            nxt = mean + std * torch.sqrt(-1 * dt) * variance_noise.float()
            nxt = nxt.to(in_dtype).float()
        if compute_log_prob:
            sv = std * torch.sqrt(-1 * dt)
            lp = -((nxt - mean) ** 2) / (2 * sv**2) - torch.log(sv) - torch.log(torch.sqrt(2 * torch.as_tensor(math.pi)))
            log_prob = lp.mean(dim=tuple(range(1, lp.ndim)))
    else:  # CPS :400-420
        std = sp * torch.sin(eta * torch.pi / 2)
        x0 = x - s * v
        x1 = x + v * (1 - s)
        mean = x0 * (1 - sp) + x1 * torch.sqrt(sp**2 - std**2)
        if nxt is None:
            nxt = mean + std * variance_noise.float()
            nxt = nxt.to(in_dtype).float()
        if compute_log_prob:
            lp = -((nxt - mean) ** 2)
            log_prob = lp.mean(dim=tuple(range(1, lp.ndim)))
    return dict(next_latents=nxt, next_latents_mean=mean, std_dev_t=std, dt=dt, log_prob=log_prob, noise_pred=v)
