"""Host-side mirror of the reference's SDE scheduler for the rollout hot path.

Same class / method / argument names and error behaviour as
`flow_factory.scheduler.FlowMatchEulerDiscreteSDEScheduler`
(reference src/flow_factory/scheduler/flow_match_euler_discrete.py:86-438, mixin contract
scheduler/abc.py:43-153), but
  * it does not depend on `diffusers`: the flow-match schedule (`set_timesteps`,
    `index_for_timestep`) is implemented here from the published algorithm;
  * `step()` runs as ONE fused HIP kernel (mi355_sde_step) on the tensors' device instead of ~25
    torch kernels and three host syncs.  CPU tensors are rejected: there is no fallback path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, fields
from typing import Any, Dict, List, Literal, Optional, Union

import numpy as np
import torch

DynamicsType = Literal["Flow-SDE", "Dance-SDE", "CPS", "ODE"]


@dataclass
class SDESchedulerOutput:
    """Single SDE step output (reference scheduler/abc.py:24-40)."""
    next_latents: Optional[torch.Tensor] = None
    next_latents_mean: Optional[torch.Tensor] = None
    std_dev_t: Optional[torch.Tensor] = None
    dt: Optional[torch.Tensor] = None
    log_prob: Optional[torch.Tensor] = None
    noise_pred: Optional[torch.Tensor] = None

    def to_dict(self) -> Dict[str, Any]:
        return {f.name: getattr(self, f.name) for f in fields(self)}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "SDESchedulerOutput":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in data.items() if k in names})


FlowMatchEulerDiscreteSDESchedulerOutput = SDESchedulerOutput


class _Config(dict):
    """Attribute + `.get` access, like a diffusers FrozenDict."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def calculate_shift(image_seq_len: int, base_seq_len: int = 256, max_seq_len: int = 4096, base_shift: float = 0.5,
                    max_shift: float = 1.15) -> float:
    """Linear interpolation of the dynamic-shift parameter mu (reference :37-47)."""
    slope = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * slope + (base_shift - slope * base_seq_len)


def set_scheduler_timesteps(scheduler, num_inference_steps: int, seq_len: Optional[int] = None, sigmas=None, device=None,
                            mu: Optional[float] = None) -> torch.Tensor:
    """Reference :49-77: sigmas = linspace(1, 1/N, N), mu from the image sequence length."""
    if sigmas is None:
        sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
    if getattr(scheduler.config, "use_flow_sigmas", False):
        sigmas = None
    if mu is None:
        if seq_len is None:
            raise AssertionError("`seq_len` must be provided if `mu` is not given.")
        cfg = scheduler.config
        mu = calculate_shift(seq_len, cfg.get("base_image_seq_len", 256), cfg.get("max_image_seq_len", 4096),
                             cfg.get("base_shift", 0.5), cfg.get("max_shift", 1.15))
    scheduler.set_timesteps(num_inference_steps if sigmas is None else None, device=device, sigmas=sigmas, mu=mu)
    return scheduler.timesteps


def host_noise_levels(scheduler, num_steps: Optional[int] = None, effective: bool = True) -> List[float]:
    """Per-step noise levels of a rollout as host floats (what `mi355_rollout` consumes: no `.item()` sync per step), derived from
    the PUBLIC SDE-scheduler contract only -- `current_sde_steps`, `noise_level`, `is_eval`, `dynamics_type` (reference
    scheduler/abc.py:76-153, flow_match_euler_discrete.py:126-198) -- so it works on the reference's own scheduler classes under
    the Flow-Factory plugin as well as on the mirrors in this package.  Equals `get_noise_level_for_timestep(t_i)` per step, with
    the `is_eval` / ODE override of `step()` (:316-317) applied -- `effective=False` leaves that override out: the SCHEDULE's value, which is
    what the reference's rollout loop hands to its callback collector (`capturable={'noise_level': ...}`, sd3_5.py:274,300) even in
    evaluation mode."""
    n = int(num_steps) if num_steps is not None else len(scheduler.timesteps)
    if effective and (bool(getattr(scheduler, "is_eval", False)) or getattr(scheduler, "dynamics_type", None) == "ODE"):
        return [0.0] * n
    cur = {int(i) for i in torch.as_tensor(scheduler.current_sde_steps).reshape(-1).tolist()}
    eta = float(scheduler.noise_level)
    return [eta if i in cur else 0.0 for i in range(n)]


def randn_tensor(shape, generator=None, device=None, dtype=None) -> torch.Tensor:
    """diffusers.utils.torch_utils.randn_tensor semantics (the draw the reference makes at sd3_5.py:242 via prepare_latents and at
    flow_match_euler_discrete.py:352): a CPU generator draws on the CPU and the result moves to `device`; a list of generators
    draws one sample each; a generator on another accelerator raises."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    shape = tuple(shape)
    gens = generator if isinstance(generator, (list, tuple)) else [generator]
    rand_device = device
    g0 = gens[0]
    if g0 is not None:
        gtype = g0.device.type
        if gtype != device.type:
            if gtype == "cpu":
                rand_device = torch.device("cpu")
            else:
                raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gtype}.")
    if isinstance(generator, (list, tuple)):
        if len(generator) == 1:
            generator = generator[0]
        else:
            if len(generator) != shape[0]:
                raise ValueError(f"got {len(generator)} generators for a batch of {shape[0]}")
            one = (1,) + shape[1:]
            return torch.cat([torch.randn(one, generator=g, device=rand_device, dtype=dtype) for g in generator], 0).to(device)
    return torch.randn(shape, generator=generator, device=rand_device, dtype=dtype).to(device)


class FlowMatchEulerDiscreteSDEScheduler:
    """Flow-match Euler scheduler with SDE noise injection on selected steps (GRPO rollouts)."""

    order = 1

    def __init__(
        self,
        noise_level: float = 0.7,
        sde_steps: Optional[Union[int, list, torch.Tensor]] = None,
        num_sde_steps: Optional[int] = None,
        seed: int = 42,
        dynamics_type: DynamicsType = "Flow-SDE",
        # diffusers FlowMatchEulerDiscreteScheduler config (SD3.5: shift=3.0, static shifting)
        num_train_timesteps: int = 1000,
        shift: float = 1.0,
        use_dynamic_shifting: bool = False,
        base_shift: Optional[float] = 0.5,
        max_shift: Optional[float] = 1.15,
        base_image_seq_len: Optional[int] = 256,
        max_image_seq_len: Optional[int] = 4096,
        time_shift_type: str = "exponential",
        shift_terminal: Optional[float] = None,
        **kwargs,
    ):
        self.config = _Config(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=use_dynamic_shifting,
                              base_shift=base_shift, max_shift=max_shift, base_image_seq_len=base_image_seq_len,
                              max_image_seq_len=max_image_seq_len, time_shift_type=time_shift_type, shift_terminal=shift_terminal,
                              **kwargs)
        self.noise_level = noise_level
        assert self.noise_level >= 0, "Noise level must be non-negative."
        self._sde_steps = torch.tensor(sde_steps, dtype=torch.int64) if sde_steps is not None else None
        self._num_sde_steps = num_sde_steps
        self.seed = seed
        self.dynamics_type = dynamics_type
        self._is_eval = False
        self._shift = shift
        # default 1000-step schedule until set_timesteps is called
        base = torch.from_numpy(np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy())
        sig = base / num_train_timesteps
        if not use_dynamic_shifting:
            sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self.sigma_min, self.sigma_max = float(sig[-1]), float(sig[0])
        self.num_inference_steps = None

    # ------------------------------------------------------------------ schedule
    @property
    def shift(self) -> float:
        return self._shift

    def time_shift(self, mu: float, sigma: float, t):
        if self.config.time_shift_type == "exponential":
            return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)
        return mu / (mu + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, sigmas=None, mu: Optional[float] = None,
                      timesteps=None) -> None:
        """Published diffusers algorithm: float32 numpy sigmas -> (dynamic | static) shift ->
        timesteps = sigma * num_train_timesteps -> append terminal sigma 0."""
        if self.config.use_dynamic_shifting and mu is None:
            raise ValueError("`mu` must be passed when `use_dynamic_shifting` is set to be `True`")
        n_train = self.config.num_train_timesteps
        if sigmas is None:
            if timesteps is None:
                timesteps = np.linspace(self.sigma_max * n_train, self.sigma_min * n_train, num_inference_steps)
            sig = np.asarray(timesteps, dtype=np.float32) / n_train
        else:
            sig = np.array(sigmas).astype(np.float32)
        if self.config.use_dynamic_shifting:
            sig = self.time_shift(mu, 1.0, sig)
        else:
            sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        if self.config.get("shift_terminal"):
            # diffusers stretch_shift_to_terminal (Qwen-Image: 0.02): rescale so that the last sigma equals shift_terminal
            one_minus = 1 - np.asarray(sig)
            sig = (1 - one_minus / (one_minus[-1] / (1 - self.config.shift_terminal))).astype(np.float32)
        sig_t = torch.from_numpy(np.asarray(sig)).to(dtype=torch.float32, device=device)
        self.timesteps = sig_t * n_train
        self.sigmas = torch.cat([sig_t, torch.zeros(1, device=sig_t.device)])
        self.num_inference_steps = len(sig_t)
        self._host_timesteps = [float(x) for x in self.timesteps.tolist()]

    def index_for_timestep(self, timestep, schedule_timesteps=None) -> int:
        sched = self.timesteps if schedule_timesteps is None else schedule_timesteps
        t = float(timestep)
        host = getattr(self, "_host_timesteps", None) if schedule_timesteps is None else None
        vals = host if host is not None else [float(x) for x in sched.tolist()]
        hits = [i for i, v in enumerate(vals) if v == t]
        if not hits:
            raise IndexError(f"timestep {t} is not on the schedule")
        return hits[1] if len(hits) > 1 else hits[0]

    # ------------------------------------------------------------------ modes
    @property
    def is_eval(self) -> bool:
        return self._is_eval

    def eval(self):
        """ODE sampling (noise_level = 0)."""
        self._is_eval = True

    def train(self, mode: bool = True):
        self._is_eval = not mode

    def rollout(self, mode: bool = True):
        self.train(mode=mode)

    def set_seed(self, seed: int):
        self.seed = seed

    # ------------------------------------------------------------------ SDE step selection (:126-198)
    @property
    def sde_steps(self) -> torch.Tensor:
        if self._sde_steps is not None:
            if not isinstance(self._sde_steps, torch.Tensor):
                self._sde_steps = torch.tensor(self._sde_steps, dtype=torch.int64)
            return self._sde_steps
        return torch.arange(0, len(self.timesteps) - 1, dtype=torch.int64)

    @property
    def num_sde_steps(self) -> int:
        return self._num_sde_steps if self._num_sde_steps is not None else len(self.sde_steps)

    @property
    def current_sde_steps(self) -> torch.Tensor:
        pool = self.sde_steps
        if self.num_sde_steps >= len(pool):
            return pool
        perm = torch.randperm(len(pool), generator=torch.Generator().manual_seed(self.seed))
        return pool[perm[: self.num_sde_steps]]

    @property
    def train_timesteps(self) -> torch.Tensor:
        return self.current_sde_steps

    def get_train_timesteps(self) -> torch.Tensor:
        return self.timesteps[self.train_timesteps]

    def get_train_sigmas(self) -> torch.Tensor:
        return self.sigmas[self.train_timesteps]

    def get_noise_levels(self) -> torch.Tensor:
        out = torch.zeros_like(self.timesteps, dtype=torch.float32)
        out[self.current_sde_steps.to(out.device)] = self.noise_level
        return out

    def host_noise_levels(self) -> List[float]:
        """Per-step noise levels as host floats (see the module-level `host_noise_levels`)."""
        return host_noise_levels(self)

    def get_noise_level_for_timestep(self, timestep):
        if not isinstance(timestep, torch.Tensor) or timestep.ndim == 0:
            idx = self.index_for_timestep(timestep)
            return self.noise_level if idx in set(self.current_sde_steps.tolist()) else 0.0
        cur = set(self.current_sde_steps.tolist())
        vals = [self.noise_level if self.index_for_timestep(t) in cur else 0.0 for t in timestep.tolist()]
        return torch.tensor(vals).to(timestep.dtype)

    def get_noise_level_for_sigma(self, sigma):
        is_scalar = not isinstance(sigma, torch.Tensor)
        s = torch.tensor([sigma], dtype=self.sigmas.dtype) if is_scalar else sigma
        sched = self.sigmas.to(s.device)
        match = s.reshape(-1, 1) == sched.reshape(1, -1)
        if not bool(match.any(dim=-1).all()):
            raise ValueError(f"Sigmas {s[~match.any(dim=-1)]} not found in scheduler sigmas.")
        idx = match.int().argmax(dim=-1)
        mask = torch.isin(idx, self.current_sde_steps.to(idx.device))
        res = torch.where(mask, torch.tensor(self.noise_level, dtype=s.dtype, device=s.device),
                          torch.tensor(0.0, dtype=s.dtype, device=s.device)).reshape(s.shape)
        return res.item() if is_scalar else res

    # ------------------------------------------------------------------ step (:243-438)
    def step(
        self,
        noise_pred: torch.Tensor,
        timestep: Union[float, torch.Tensor],
        latents: torch.Tensor,
        next_latents: Optional[torch.Tensor] = None,
        timestep_next: Optional[Union[float, torch.Tensor]] = None,
        generator: Optional[torch.Generator] = None,
        noise_level: Optional[Union[int, float, torch.Tensor]] = None,
        compute_log_prob: bool = True,
        return_dict: bool = True,
        return_kwargs: List[str] = ["next_latents", "next_latents_mean", "std_dev_t", "dt", "log_prob", "noise_pred"],
        dynamics_type: Optional[DynamicsType] = None,
        sigma_max: Optional[float] = None,
        variance_noise: Optional[torch.Tensor] = None,
    ):
        """One ODE/SDE step + Gaussian log-prob, as a single fused HIP kernel.

        `variance_noise` (extension): pre-drawn eps; otherwise eps is drawn here with
        `torch.randn(noise_pred.shape, generator=generator, device=..., dtype=float32)` -- the same
        call, shape, dtype and generator the reference makes (:352-357), so seeds reproduce."""
        from . import engine

        if not latents.is_cuda:
            raise RuntimeError("mi355_flow scheduler.step: tensors must live on the GPU (no CPU fallback on the hot path)")
        dev = latents.device
        B = latents.shape[0]
        if timestep_next is None:
            if isinstance(timestep, int) or (isinstance(timestep, torch.Tensor) and not timestep.is_floating_point()):
                idxs = [int(timestep)]
            elif isinstance(timestep, torch.Tensor):
                if timestep.ndim == 0:
                    idxs = [self.index_for_timestep(timestep)]
                elif timestep.ndim == 1:
                    idxs = [self.index_for_timestep(t) for t in timestep]
                else:
                    raise ValueError(
                        f"`timestep` must be a scalar or 1D tensor, got shape {tuple(timestep.shape)}. "
                        f"If using expanded timesteps (e.g. for Wan models), pass the original scalar timestep `t` instead.")
            elif isinstance(timestep, float):
                idxs = [self.index_for_timestep(timestep)]
            else:
                raise TypeError(f"`timestep` must be float, or torch.Tensor, got {type(timestep).__name__}.")
            sigma = self.sigmas[idxs].to(dev)
            sigma_prev = self.sigmas[[i + 1 for i in idxs]].to(dev)
        else:
            # exact fp32 quotient (GPU tensor/scalar division multiplies by a rounded reciprocal)
            sigma = (torch.as_tensor(timestep, dtype=torch.float32, device=dev).double() / 1000).float()
            sigma_prev = (torch.as_tensor(timestep_next, dtype=torch.float32, device=dev).double() / 1000).float()
        dyn = dynamics_type or self.dynamics_type
        if dyn not in ("Flow-SDE", "Dance-SDE", "CPS", "ODE"):
            raise ValueError(f"unknown dynamics_type {dyn!r}")
        if self.is_eval or dyn == "ODE":
            noise_level = 0.0
        elif noise_level is None:
            noise_level = self.get_noise_level_for_sigma(sigma)
        if sigma_max is None:
            sigma_max = getattr(self, "_host_sigma1", None)
            if sigma_max is None or getattr(self, "_host_sigma1_n", None) != self.num_inference_steps:
                sigma_max = float(self.sigmas[1])
                self._host_sigma1, self._host_sigma1_n = sigma_max, self.num_inference_steps
        if next_latents is None and dyn != "ODE" and variance_noise is None:
            variance_noise = randn_tensor(noise_pred.shape, generator=generator, device=noise_pred.device, dtype=torch.float32)
        want = [k for k in return_kwargs if k in ("next_latents", "next_latents_mean", "std_dev_t", "dt", "noise_pred")]
        if not return_dict:
            want = ["next_latents", "next_latents_mean", "std_dev_t", "dt", "noise_pred"]
        o = engine.sde_step(noise_pred, None, 1.0, latents, sigma, sigma_prev, noise_level, float(sigma_max), dyn,
                            noise=variance_noise, next_latents=next_latents, compute_log_prob=compute_log_prob, want=want)
        view = (-1,) + (1,) * (latents.dim() - 1)
        res = dict(
            next_latents=o.next_latents if next_latents is None else next_latents.float(),
            next_latents_mean=o.next_latents_mean,
            std_dev_t=o.std_dev_t.view(view) if o.std_dev_t is not None else None,
            dt=o.dt.view(view) if o.dt is not None else None,
            log_prob=o.log_prob if compute_log_prob else None,
            noise_pred=o.noise_pred,
        )
        if not return_dict:
            return (res["next_latents"], res["next_latents_mean"], res["noise_pred"], res["log_prob"], res["std_dev_t"], res["dt"])
        return SDESchedulerOutput.from_dict({k: res[k] for k in return_kwargs if k in res})
