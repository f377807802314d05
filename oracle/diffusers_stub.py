"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Minimal stand-in for the four `diffusers` symbols that the reference's SDE
scheduler imports (reference: src/flow_factory/scheduler/flow_match_euler_discrete.py:25-28
and src/flow_factory/scheduler/abc.py:21).  `diffusers` (pyproject.toml:34,
`diffusers>=0.36.0`; the vendored submodule in /root/reference/diffusers is
empty, no SHA recoverable) is not installed in this image, so the published
algorithm of these symbols is restated here:

  * `BaseOutput`                      -- dataclass-ish ordered container.
  * `randn_tensor`                    -- torch.randn on `device` with `generator`.
  * `retrieve_timesteps`              -- calls scheduler.set_timesteps(sigmas=..., mu=...).
  * `FlowMatchEulerDiscreteScheduler` -- __init__/set_timesteps/index_for_timestep/time_shift.

With this stub installed in `sys.modules` the reference's OWN
`FlowMatchEulerDiscreteSDEScheduler` (step(), SDE-step selection, ...) runs
unmodified; see `ref_loader.py`.  Only the schedule construction
(`set_timesteps`) is restated third-party arithmetic.
"""
from __future__ import annotations

import inspect
import math
import sys
import types
from typing import Any, List, Optional, Union

import numpy as np
import torch


class BaseOutput:
    """Restatement of diffusers.utils.outputs.BaseOutput: the reference only needs a
    dataclass base with attribute access (`output.next_latents`) and key access."""

    def __getitem__(self, k):
        import dataclasses

        if isinstance(k, str):
            return getattr(self, k)
        return self.to_tuple()[k]

    def keys(self):
        import dataclasses

        return [f.name for f in dataclasses.fields(self) if getattr(self, f.name) is not None]

    def to_tuple(self):
        return tuple(getattr(self, k) for k in self.keys())


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """Restatement of diffusers.utils.torch_utils.randn_tensor (single-generator case)."""
    layout = layout or torch.strided
    device = device or torch.device("cpu")
    if isinstance(generator, list):
        shape = (1,) + tuple(shape[1:])
        latents = [
            torch.randn(shape, generator=generator[i], device=device, dtype=dtype, layout=layout)
            for i in range(len(generator))
        ]
        return torch.cat(latents, dim=0)
    return torch.randn(tuple(shape), generator=generator, device=device, dtype=dtype, layout=layout)


class _FrozenConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # hasattr() must see AttributeError
            raise AttributeError(k) from e


class FlowMatchEulerDiscreteScheduler:
    """Restatement of diffusers' FlowMatchEulerDiscreteScheduler (schedule construction only).

    SD3.5-medium ships `scheduler_config.json` with shift=3.0,
    use_dynamic_shifting=False, num_train_timesteps=1000 (from memory; the
    checkpoint is not in this image).
    """

    order = 1

    def __init__(
        self,
        num_train_timesteps: int = 1000,
        shift: float = 1.0,
        use_dynamic_shifting: bool = False,
        base_shift: Optional[float] = 0.5,
        max_shift: Optional[float] = 1.15,
        base_image_seq_len: Optional[int] = 256,
        max_image_seq_len: Optional[int] = 4096,
        invert_sigmas: bool = False,
        shift_terminal: Optional[float] = None,
        use_karras_sigmas: Optional[bool] = False,
        use_exponential_sigmas: Optional[bool] = False,
        use_beta_sigmas: Optional[bool] = False,
        time_shift_type: str = "exponential",
        stochastic_sampling: bool = False,
        **unused,
    ):
        self.config = _FrozenConfig(
            num_train_timesteps=num_train_timesteps,
            shift=shift,
            use_dynamic_shifting=use_dynamic_shifting,
            base_shift=base_shift,
            max_shift=max_shift,
            base_image_seq_len=base_image_seq_len,
            max_image_seq_len=max_image_seq_len,
            invert_sigmas=invert_sigmas,
            shift_terminal=shift_terminal,
            use_karras_sigmas=use_karras_sigmas,
            use_exponential_sigmas=use_exponential_sigmas,
            use_beta_sigmas=use_beta_sigmas,
            time_shift_type=time_shift_type,
            stochastic_sampling=stochastic_sampling,
        )
        timesteps = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        timesteps = torch.from_numpy(timesteps).to(dtype=torch.float32)
        sigmas = timesteps / num_train_timesteps
        if not use_dynamic_shifting:
            sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.timesteps = sigmas * num_train_timesteps
        self._step_index = None
        self._begin_index = None
        self._shift = shift
        self.sigmas = sigmas.to("cpu")
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()

    @property
    def shift(self):
        return self._shift

    def _sigma_to_t(self, sigma):
        return sigma * self.config.num_train_timesteps

    def time_shift(self, mu: float, sigma: float, t):
        if self.config.time_shift_type == "exponential":
            return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)
        return mu / (mu + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, timesteps=None):
        if self.config.use_dynamic_shifting and mu is None:
            raise ValueError("`mu` must be passed when `use_dynamic_shifting` is set to be `True`")
        is_timesteps_provided = timesteps is not None
        if is_timesteps_provided:
            timesteps = np.array(timesteps).astype(np.float32)
        if sigmas is None:
            if timesteps is None:
                timesteps = np.linspace(
                    self._sigma_to_t(self.sigma_max), self._sigma_to_t(self.sigma_min), num_inference_steps
                )
            sigmas = timesteps / self.config.num_train_timesteps
        else:
            sigmas = np.array(sigmas).astype(np.float32)
            num_inference_steps = len(sigmas)
        if self.config.use_dynamic_shifting:
            sigmas = self.time_shift(mu, 1.0, sigmas)
        else:
            sigmas = self.shift * sigmas / (1 + (self.shift - 1) * sigmas)
        sigmas = torch.from_numpy(np.asarray(sigmas)).to(dtype=torch.float32, device=device)
        if not is_timesteps_provided:
            timesteps = sigmas * self.config.num_train_timesteps
        else:
            timesteps = torch.from_numpy(timesteps).to(dtype=torch.float32, device=device)
        sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self.num_inference_steps = num_inference_steps
        self.timesteps = timesteps
        self.sigmas = sigmas
        self._step_index = None
        self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        if schedule_timesteps is None:
            schedule_timesteps = self.timesteps
        indices = (schedule_timesteps == timestep).nonzero()
        pos = 1 if len(indices) > 1 else 0
        return indices[pos].item()


class UniPCMultistepScheduler:
    """Restatement of diffusers' UniPCMultistepScheduler for the part the GRPO rollout uses: schedule construction with
    `use_flow_sigmas=True` (the Wan pipelines) and `index_for_timestep`.  The multistep predictor-corrector `step()` (only reached by the
    reference in evaluation mode, scheduler/unipc_multistep.py:282-285) delegates to oracle/unipc_ref.py -- the published algorithm restated
    tensor by tensor, PARITY UNPINNED (see that file's header): with it the reference's own adapter loop runs in evaluation mode on the CPU,
    which pins the plugin's CONTROL FLOW (RNG order, cast_latents, expert / guidance per step) against it, not the solver body."""

    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 use_flow_sigmas: bool = True, flow_shift: float = 3.0, final_sigmas_type: str = "zero", solver_type: str = "bh2",
                 predict_x0: bool = True, lower_order_final: bool = True, disable_corrector=(), thresholding: bool = False, **unused):
        if not use_flow_sigmas:
            raise NotImplementedError("diffusers_stub.UniPCMultistepScheduler: only the use_flow_sigmas schedule is restated")
        self.config = _FrozenConfig(num_train_timesteps=num_train_timesteps, solver_order=solver_order, prediction_type=prediction_type,
                                    use_flow_sigmas=use_flow_sigmas, flow_shift=flow_shift, final_sigmas_type=final_sigmas_type,
                                    solver_type=solver_type, predict_x0=predict_x0, lower_order_final=lower_order_final,
                                    disable_corrector=list(disable_corrector), thresholding=thresholding, solver_p=None)
        self._solver = None
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32)[::-1].copy())
        self.sigmas = torch.zeros(num_train_timesteps + 1)
        self.num_inference_steps = None
        self._step_index = None
        self._begin_index = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        n_train, shift = self.config.num_train_timesteps, self.config.flow_shift
        alphas = np.linspace(1, 1 / n_train, num_inference_steps + 1)
        sigmas = 1.0 - alphas
        sigmas = np.flip(shift * sigmas / (1 + (shift - 1) * sigmas))[:-1].copy()
        timesteps = (sigmas * n_train).copy()
        sigma_last = sigmas[-1] if self.config.final_sigmas_type == "sigma_min" else 0.0
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [sigma_last]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self._step_index = None
        self._begin_index = None
        self._solver = None                   # (set_timesteps resets model_outputs / lower_order_nums / last_sample)

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        if schedule_timesteps is None:
            schedule_timesteps = self.timesteps
        indices = (schedule_timesteps == timestep).nonzero()
        if len(indices) == 0:
            return len(self.timesteps) - 1
        return indices[1 if len(indices) > 1 else 0].item()

    def step(self, model_output, timestep, sample, return_dict: bool = True):
        from .unipc_ref import UniPCRef
        if self._solver is None:
            c = self.config
            self._solver = UniPCRef(self.sigmas.tolist(), solver_order=c.solver_order, solver_type=c.solver_type,
                                    lower_order_final=c.lower_order_final, disable_corrector=c.disable_corrector)
        prev = self._solver.step(model_output, sample)
        return (prev,)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    """Restatement of diffusers' pipeline helper `retrieve_timesteps` (sigmas branch)."""
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed.")
    if sigmas is not None:
        if "sigmas" not in set(inspect.signature(scheduler.set_timesteps).parameters.keys()):
            raise ValueError("scheduler does not accept custom sigmas")
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
        timesteps = scheduler.timesteps
        num_inference_steps = len(timesteps)
    elif timesteps is not None:
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
        timesteps = scheduler.timesteps
        num_inference_steps = len(timesteps)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
        timesteps = scheduler.timesteps
    return timesteps, num_inference_steps


def install() -> None:
    """Register the stub under the `diffusers.*` module names the reference imports."""

    def mod(name: str) -> types.ModuleType:
        m = types.ModuleType(name)
        m.__path__ = []  # behave like a package
        sys.modules[name] = m
        return m

    if "diffusers" in sys.modules and not getattr(sys.modules["diffusers"], "__mi355_stub__", False):
        return  # a real diffusers is importable: use it
    root = mod("diffusers")
    root.__mi355_stub__ = True
    utils = mod("diffusers.utils")
    outputs = mod("diffusers.utils.outputs")
    outputs.BaseOutput = BaseOutput
    tu = mod("diffusers.utils.torch_utils")
    tu.randn_tensor = randn_tensor
    utils.outputs, utils.torch_utils = outputs, tu
    mod("diffusers.pipelines")
    mod("diffusers.pipelines.stable_diffusion_3")
    p = mod("diffusers.pipelines.stable_diffusion_3.pipeline_stable_diffusion_3")
    p.retrieve_timesteps = retrieve_timesteps
    mod("diffusers.schedulers")
    s = mod("diffusers.schedulers.scheduling_flow_match_euler_discrete")
    s.FlowMatchEulerDiscreteScheduler = FlowMatchEulerDiscreteScheduler
