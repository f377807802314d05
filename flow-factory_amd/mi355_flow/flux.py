"""FLUX.1 rollout on the native engine (mi355_flux_*): host mirror of `Flux1Adapter.inference` / `.forward`
(reference src/flow_factory/models/flux/flux1.py:151-289, :294-346) -- SURVEY.md 8(f) row N3.

Same contract as the SD3.5 path (`mi355_flow.adapter`): reference argument names and defaults, the reference's RNG draw
order, trajectory / log-prob / callback collectors, no CPU or PyTorch fallback.  FLUX specifics kept from the reference:
packed latents `(B, h/2*w/2, 64)`, `img_ids` from `prepare_latents`, `timestep = t / 1000`, embedded guidance (no CFG,
no negative prompt), dynamic-shift `mu` from the image sequence length.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from ._lib import DYNAMICS, FluxCfg
from .engine import WeightHolder, _bf16c, _ptr, _stream, dtype_code, sde_step
from .samples import Flux1Sample
from .scheduler import (FlowMatchEulerDiscreteSDEScheduler, SDESchedulerOutput, host_noise_levels, randn_tensor,
                        set_scheduler_timesteps)
from .trajectory import TrajectoryIndicesType, _resolve, collect_rollout

_DTYPE_MAP = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16,
              "fp32": torch.float32, "float32": torch.float32}
VAE_SCALE_FACTOR = 8


@dataclass
class FluxConfig:
    """diffusers FluxTransformer2DModel config fields the engine needs (FLUX.1-dev defaults)."""
    in_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    num_attention_heads: int = 24
    attention_head_dim: int = 128
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)
    time_proj_dim: int = 256
    eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    def to_c(self) -> FluxCfg:
        return FluxCfg(self.in_channels, self.num_layers, self.num_single_layers, self.num_attention_heads, self.attention_head_dim,
                       self.joint_attention_dim, self.pooled_projection_dim, int(self.guidance_embeds), self.time_proj_dim,
                       (C.c_int32 * 3)(*self.axes_dims_rope), self.eps)


def pack_latents(lat: torch.Tensor) -> torch.Tensor:
    """FluxPipeline._pack_latents: (B, C, h, w) -> (B, h/2*w/2, 4C)."""
    B, Cc, h, w = lat.shape
    return lat.view(B, Cc, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, (h // 2) * (w // 2), Cc * 4)


def unpack_latents(x: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """FluxPipeline._unpack_latents on the latent grid: (B, h/2*w/2, 4C) -> (B, C, h, w)."""
    B, _, ch = x.shape
    return x.view(B, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(B, ch // 4, h, w)


def prepare_latent_image_ids(hp: int, wp: int, device, dtype) -> torch.Tensor:
    """FluxPipeline._prepare_latent_image_ids: (hp*wp, 3) = [0, row, col]."""
    ids = torch.zeros(hp, wp, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(hp)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(wp)[None, :]
    return ids.reshape(hp * wp, 3).to(device=device, dtype=dtype)


def model_scalar(value: float, dtype: torch.dtype) -> float:
    """`x.to(hidden_states.dtype) * 1000` of FluxTransformer2DModel.forward for a python / fp32 scalar."""
    return float((torch.tensor(float(value), dtype=torch.float32).to(dtype) * 1000).float())


class FluxEngine(WeightHolder):
    """Owns the packed bf16 copy of the FLUX transformer weights (mi355_flux)."""

    def __init__(self, cfg: FluxConfig = FluxConfig()):
        self.lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        c = cfg.to_c()
        _lib.check(self.lib.mi355_flux_create(C.byref(c), C.byref(h)), "flux_create")
        self._h = h
        self._plans: Dict[tuple, "FluxPlan"] = {}

    _ABI, _WHAT = "flux", "FLUX transformer"

    def plan(self, batch: int, latent_h: int, latent_w: int, n_text: int, max_steps: int) -> "FluxPlan":
        key = (batch, latent_h, latent_w, n_text)
        p = self._plans.get(key)
        if p is None or p.max_steps < max_steps:
            if p is not None:
                p.close()
            p = FluxPlan(self, batch, latent_h, latent_w, n_text, max_steps)
            self._plans[key] = p
        return p

    # ---------------------------------------------------------------- weight gradients (mi355_flow/autograd.py: flux_replay)
    def grad_supported(self, name: str) -> int:
        """1 = the native backward produces a gradient for this parameter (the linear layers inside the transformer blocks), 0 = it does not."""
        return 1 if self.lib.mi355_flux_grad_supported(self._h, name.encode()) == 0 else 0

    def set_grad(self, name: str, grad: torch.Tensor) -> None:
        """Register the buffer the next backward writes d loss / d `name` into (same shape as the parameter): fp32, or bf16 for the weights /
        biases of the linear layers inside the blocks (`grad_supported(name) == 1`) -- the engine then rounds its fp32 sums to bf16 itself."""
        if grad.dtype not in (torch.float32, torch.bfloat16) or not grad.is_contiguous():
            raise ValueError("mi355_flow: gradient buffers are contiguous fp32 (or bf16) tensors")
        _lib.check(self.lib.mi355_flux_set_grad_typed(self._h, name.encode(), _ptr(grad), dtype_code(grad.dtype)), f"flux_set_grad({name})")

    def clear_grads(self) -> None:
        _lib.check(self.lib.mi355_flux_clear_grads(self._h), "flux_clear_grads")

    def close(self) -> None:
        for p in self._plans.values():
            p.close()
        self._plans.clear()
        if self._h:
            self.lib.mi355_flux_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FluxPlan:
    def __init__(self, engine: FluxEngine, batch: int, latent_h: int, latent_w: int, n_text: int, max_steps: int):
        self.engine, self.lib = engine, engine.lib
        self.batch, self.h, self.w, self.n_text, self.max_steps = batch, latent_h, latent_w, n_text, max_steps
        self.Ni = (latent_h // 2) * (latent_w // 2)
        self.C = engine.cfg.in_channels
        h = C.c_void_p()
        _lib.check(self.lib.mi355_flux_plan_create(engine._h, batch, latent_h, latent_w, n_text, max_steps, C.byref(h)), "flux_plan_create")
        self._h = h

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.mi355_flux_plan_workspace_bytes(self._h))

    def close(self) -> None:
        if self._h:
            self.lib.mi355_flux_plan_destroy(self._h)
            self._h = None

    def transformer_forward(self, latents: torch.Tensor, t_model: torch.Tensor, guidance_model: Optional[torch.Tensor],
                            prompt_embeds: torch.Tensor, pooled: torch.Tensor) -> torch.Tensor:
        """latents (B, Ni, 64) packed; t_model / guidance_model (B,) = the values the network embeds."""
        B = self.batch
        assert latents.shape == (B, self.Ni, self.C), latents.shape
        dev = latents.device
        tm = t_model.to(device=dev, dtype=torch.float32).reshape(-1)
        tm = (tm.expand(B) if tm.numel() == 1 else tm).contiguous()
        gm = None
        if guidance_model is not None:
            gm = guidance_model.to(device=dev, dtype=torch.float32).reshape(-1)
            gm = (gm.expand(B) if gm.numel() == 1 else gm).contiguous()
        out = torch.empty((B, self.Ni, self.C), device=dev, dtype=torch.bfloat16)
        latents = latents.contiguous()
        pe, pp = _bf16c(prompt_embeds), _bf16c(pooled)
        _lib.check(self.lib.mi355_flux_forward(self._h, _stream(), _ptr(latents), dtype_code(latents.dtype), _ptr(tm), _ptr(gm), _ptr(pe),
                                               _ptr(pp), _ptr(out)), "flux_forward")
        return out

    def _model_scalars(self, latents, t_model, guidance_model):
        B, dev = self.batch, latents.device
        tm = t_model.to(device=dev, dtype=torch.float32).reshape(-1)
        tm = (tm.expand(B) if tm.numel() == 1 else tm).contiguous()
        gm = None
        if guidance_model is not None:
            gm = guidance_model.to(device=dev, dtype=torch.float32).reshape(-1)
            gm = (gm.expand(B) if gm.numel() == 1 else gm).contiguous()
        return tm, gm

    # ---------------------------------------------------------------- differentiable forward (optimize() replay)
    def forward_train(self, latents: torch.Tensor, t_model: torch.Tensor, guidance_model: Optional[torch.Tensor],
                      prompt_embeds: torch.Tensor, pooled: torch.Tensor) -> torch.Tensor:
        """mi355_flux_forward_train: `transformer_forward` on per-block activation buffers -- the same kernel binaries, so the velocity is
        bit-identical -- keeping what `backward` needs in the plan's training stash (ONE per plan; every call takes a serial number)."""
        assert latents.shape == (self.batch, self.Ni, self.C), latents.shape
        tm, gm = self._model_scalars(latents, t_model, guidance_model)
        out = torch.empty((self.batch, self.Ni, self.C), device=latents.device, dtype=torch.bfloat16)
        latents = latents.contiguous()
        pe, pp = _bf16c(prompt_embeds), _bf16c(pooled)
        _lib.check(self.lib.mi355_flux_forward_train(self._h, _stream(), _ptr(latents), dtype_code(latents.dtype), _ptr(tm), _ptr(gm), _ptr(pe),
                                                     _ptr(pp), _ptr(out)), "flux_forward_train")
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        return out

    def backward(self, dv: torch.Tensor) -> None:
        """mi355_flux_backward: d loss / d v [B, Ni, C] fp32 of the LAST `forward_train` -> the buffers registered with `FluxEngine.set_grad`."""
        dv = dv.to(torch.float32).contiguous()
        assert dv.shape == (self.batch, self.Ni, self.C), dv.shape
        _lib.check(self.lib.mi355_flux_backward(self._h, _stream(), _ptr(dv)), "flux_backward")

    @property
    def training_bytes(self) -> int:
        return int(self.lib.mi355_flux_plan_training_bytes(self._h))

    def rollout(self, timesteps: Sequence[float], sigmas: Sequence[float], noise_levels: Sequence[float], dynamics: str,
                guidance_scale: float, init_latents: torch.Tensor, storage_dtype: torch.dtype, step_noise: Optional[torch.Tensor],
                prompt_embeds: torch.Tensor, pooled: torch.Tensor, keep_positions: Optional[Sequence[int]] = None,
                compute_log_prob: bool = True):
        """Returns (kept_latents [n_kept, B, Ni, 64] storage dtype, log_probs [N, B] fp32 (nan where not computed), final)."""
        N, B = len(timesteps), self.batch
        assert len(sigmas) == N + 1 and len(noise_levels) == N
        dev = init_latents.device
        keep = list(range(N + 1)) if keep_positions is None else sorted(set(int(k) for k in keep_positions))
        slots = [-1] * (N + 1)
        for s, pos in enumerate(keep):
            slots[pos] = s
        shape = (B, self.Ni, self.C)
        assert tuple(init_latents.shape) == shape, init_latents.shape
        out_lat = torch.empty((len(keep),) + shape, device=dev, dtype=storage_dtype)
        out_lp = torch.full((N, B), float("nan"), device=dev, dtype=torch.float32)
        out_fin = torch.empty(shape, device=dev, dtype=storage_dtype)
        fa = C.c_float * N
        ts_c, nl_c = fa(*[float(t) for t in timesteps]), fa(*[float(e) for e in noise_levels])
        sg_c = (C.c_float * (N + 1))(*[float(s) for s in sigmas])
        sl_c = (C.c_int32 * (N + 1))(*slots)
        init_latents = init_latents.contiguous()
        if step_noise is not None:
            step_noise = step_noise.contiguous()
            assert step_noise.dtype == torch.float32 and step_noise.shape == (N,) + shape, step_noise.shape
        pe, pp = _bf16c(prompt_embeds), _bf16c(pooled)
        _lib.check(self.lib.mi355_flux_rollout(
            self._h, _stream(), N, ts_c, sg_c, nl_c, DYNAMICS[dynamics], float(guidance_scale), _ptr(init_latents),
            dtype_code(init_latents.dtype), dtype_code(storage_dtype), _ptr(step_noise), _ptr(pe), _ptr(pp), sl_c, _ptr(out_lat),
            _ptr(out_lp), _ptr(out_fin), int(bool(compute_log_prob))), "flux_rollout")
        return out_lat, out_lp, out_fin


class FluxRolloutMixin:
    """`inference()` / `forward()` of `Flux1Adapter` (reference models/flux/flux1.py:151-289, :294-346) on the engine.  Host classes
    provide `engine` (FluxEngine), `scheduler`, `device`, `latent_storage_dtype`, `encode_prompt`, `decode_latents(latents, height,
    width, output_type)`.  Used by the standalone `Flux1NativeAdapter` below and, mixed in FRONT of the reference's own `Flux1Adapter`,
    by `mi355_flow.flow_factory_plugin.Flux1NativeAdapter`."""

    _sample_cls = Flux1Sample
    _output_cls = SDESchedulerOutput
    _set_timesteps = staticmethod(set_scheduler_timesteps)

    def _before_engine_call(self) -> None:
        """Hook run at the top of inference() / forward(): the Flow-Factory plugin re-binds changed weights here."""

    def _check_joint_attention_kwargs(self, jak) -> None:
        jak = dict(jak or {})
        scale = float(jak.pop("scale", 1.0))
        if jak:
            raise NotImplementedError(f"mi355_flow: joint_attention_kwargs {sorted(jak)} are not supported by the native engine "
                                      "(only the LoRA `scale` is)")
        live = getattr(self, "_live_weights", None)
        if live is not None:
            live.set_lora_scale(scale)

    def cast_latents(self, latents: torch.Tensor, default_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        target = self.latent_storage_dtype or default_dtype
        if target is None or latents.dtype == target:
            return latents
        if target == torch.float16:
            latents = latents.clamp(-65504.0, 65504.0)
        return latents.to(target)

    # ------------------------------------------------------------------ rollout (flux1.py:151-289)
    @torch.no_grad()
    def inference(
        self,
        prompt: Optional[Union[str, List[str]]] = None,
        height: int = 512,
        width: int = 512,
        num_inference_steps: int = 28,
        guidance_scale: float = 3.5,
        generator: Optional[torch.Generator] = None,
        prompt_ids: Optional[torch.Tensor] = None,
        prompt_embeds: Optional[torch.Tensor] = None,
        pooled_prompt_embeds: Optional[torch.Tensor] = None,
        joint_attention_kwargs: Optional[Dict[str, Any]] = None,
        compute_log_prob: bool = True,
        extra_call_back_kwargs: List[str] = [],
        trajectory_indices: TrajectoryIndicesType = "all",
    ) -> List[Flux1Sample]:
        self._before_engine_call()
        device = self.device
        self._check_joint_attention_kwargs(joint_attention_kwargs)
        if prompt_embeds is None:
            enc = self.encode_prompt(prompt)
            prompt_embeds, pooled_prompt_embeds, prompt_ids = enc["prompt_embeds"], enc["pooled_prompt_embeds"], enc["prompt_ids"]
        else:
            prompt_embeds, pooled_prompt_embeds = prompt_embeds.to(device), pooled_prompt_embeds.to(device)
        B = len(prompt_embeds)
        dtype = prompt_embeds.dtype
        Cl = self.engine.cfg.in_channels // 4
        # FluxPipeline.prepare_latents: height = 2 * (height // (vae_scale_factor * 2))
        h = 2 * (int(height) // (VAE_SCALE_FACTOR * 2))
        w = 2 * (int(width) // (VAE_SCALE_FACTOR * 2))
        N = int(num_inference_steps)
        Ni = (h // 2) * (w // 2)

        # RNG in the reference's order: prepare_latents draws (B, 16, h, w) in the prompt dtype and packs it; the scheduler then
        # draws one fp32 tensor of the PACKED shape per step (randn_tensor(noise_pred.shape)), also when noise_level == 0
        # (no step draws at all under ODE dynamics: the reference's ODE branch draws nothing)
        dyn = self.scheduler.dynamics_type
        latents = pack_latents(randn_tensor((B, Cl, h, w), generator=generator, device=device, dtype=dtype))
        latent_image_ids = prepare_latent_image_ids(h // 2, w // 2, device, dtype)
        step_noise = None
        if dyn != "ODE":
            step_noise = torch.empty((N, B, Ni, Cl * 4), device=device, dtype=torch.float32)
            for i in range(N):
                step_noise[i] = randn_tensor((B, Ni, Cl * 4), generator=None, device=device, dtype=torch.float32)

        timesteps = self._set_timesteps(self.scheduler, N, seq_len=latents.shape[1], device=device)
        ts_host = [float(t) for t in timesteps.tolist()]
        sig_host = [float(s) for s in self.scheduler.sigmas.tolist()]
        eta_host = host_noise_levels(self.scheduler, N)
        storage = self.latent_storage_dtype or dtype
        plan = self.engine.plan(B, h, w, prompt_embeds.shape[1], N)
        stepwise = any(k != "noise_level" for k in extra_call_back_kwargs)
        kept = _resolve(trajectory_indices, N + 1)
        keep_positions = list(range(N + 1)) if kept is None else sorted(kept)
        step_outputs = None
        if not stepwise:
            lat_kept, log_probs, final = plan.rollout(ts_host, sig_host, eta_host, self.scheduler.dynamics_type, guidance_scale, latents,
                                                      storage, step_noise, prompt_embeds, pooled_prompt_embeds,
                                                      keep_positions=keep_positions, compute_log_prob=compute_log_prob)
            pos_to_slot = {p: s for s, p in enumerate(keep_positions)}
        else:
            # per-step engine calls (still all-HIP) for rollouts that ask for per-step callback tensors
            lat_kept, log_probs, step_outputs = self._rollout_stepwise(plan, ts_host, sig_host, eta_host, guidance_scale, latents, storage,
                                                                       step_noise, prompt_embeds, pooled_prompt_embeds, compute_log_prob,
                                                                       extra_call_back_kwargs)
            final = lat_kept[N]
            pos_to_slot = {p: p for p in range(N + 1)}

        traj = collect_rollout(trajectory_indices, N, lambda pos: lat_kept[pos_to_slot[pos]], log_probs, eta_host, compute_log_prob,
                               step_outputs, extra_call_back_kwargs,
                               captured_noise_levels=host_noise_levels(self.scheduler, N, effective=False), dynamics=dyn)
        images = self.decode_latents(final, height, width, output_type="pt")
        return [
            self._sample_cls(
                timesteps=timesteps,
                **traj.per_sample(b),
                prompt=prompt[b] if isinstance(prompt, list) else prompt,
                prompt_ids=prompt_ids[b] if prompt_ids is not None else None,
                prompt_embeds=prompt_embeds[b],
                pooled_prompt_embeds=pooled_prompt_embeds[b],
                height=height, width=width,
                image=images[b] if images is not None else None,
                img_ids=latent_image_ids,
            )
            for b in range(B)
        ]

    def _rollout_stepwise(self, plan, ts, sig, eta, guidance, latents, storage, step_noise, pe, pp, compute_log_prob, extra_keys):
        N, B = len(ts), latents.shape[0]
        cur = self.cast_latents(latents, storage)
        all_lat = [cur]
        log_probs = torch.full((N, B), float("nan"), device=latents.device)
        outs = []
        want = tuple(k for k in extra_keys if k in ("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt"))
        f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
        gm = (torch.tensor(float(guidance)).to(storage) * 1000).float().reshape(1) if self.engine.cfg.guidance_embeds else None
        for i in range(N):
            t_next = ts[i + 1] if i + 1 < N else 0.0
            clp = compute_log_prob and eta[i] > 0
            tm = ((f32(ts[i]) / f32(1000.0)).to(storage) * 1000).float().reshape(1)      # as mi355_flux_rollout's host math
            v = plan.transformer_forward(cur, tm, gm, pe, pp)
            o = sde_step(v, None, 1.0, cur, f32(ts[i]) / f32(1000.0), f32(t_next) / f32(1000.0), eta[i], sig[1], self.scheduler.dynamics_type,
                         noise=step_noise[i] if step_noise is not None else None, compute_log_prob=clp, want=want)
            if clp:
                log_probs[i] = o.log_prob
            cur = o.next_storage
            all_lat.append(cur)
            outs.append(o)
        return all_lat, log_probs, outs

    # ------------------------------------------------------------------ single step / replay (flux1.py:294-346)
    def _grad_fallback(self, why: str, kwargs: Dict[str, Any]):
        """Grad-mode forward() the native backward cannot serve.  Standalone: there is no other implementation -- raise (the Flow-Factory
        plugin overrides this with the reference's autograd path)."""
        raise NotImplementedError(f"mi355_flow: FLUX.1 forward() with autograd is not available natively: {why}")

    def forward(
        self,
        t: torch.Tensor,
        latents: torch.Tensor,
        prompt_embeds: torch.Tensor,
        pooled_prompt_embeds: torch.Tensor,
        img_ids: Optional[torch.Tensor] = None,
        t_next: Optional[torch.Tensor] = None,
        next_latents: Optional[torch.Tensor] = None,
        guidance_scale: Union[float, List[float]] = 3.5,
        noise_level: Optional[float] = None,
        joint_attention_kwargs: Optional[Dict[str, Any]] = None,
        compute_log_prob: bool = True,
        return_kwargs: List[str] = ["noise_pred", "next_latents", "next_latents_mean", "std_dev_t", "dt", "log_prob"],
        height: Optional[int] = None,
        width: Optional[int] = None,
    ) -> SDESchedulerOutput:
        kw = dict(t=t, latents=latents, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds, img_ids=img_ids, t_next=t_next,
                  next_latents=next_latents, guidance_scale=guidance_scale, noise_level=noise_level, joint_attention_kwargs=joint_attention_kwargs,
                  compute_log_prob=compute_log_prob, return_kwargs=return_kwargs, height=height, width=width)
        if torch.is_grad_enabled() and getattr(self, "_live_weights", None) is not None:
            # optimize() (trainers/grpo.py:263): the replay of a stored transition WITH autograd on the engine's differentiable forward +
            # native backward (mi355_flow.autograd.flux_replay) when its backward covers the trainable set; the matching-loss trainers'
            # forward without a stored transition (noise_pred only) runs the same path with the log-prob switched off
            from . import autograd as AG
            self._before_engine_call()
            why = AG.unsupported_reason(self)
            sampled = next_latents is None and ("next_latents" in return_kwargs or (compute_log_prob and "log_prob" in return_kwargs))
            if why is None and sampled:
                why = "a sampled next state (or its log-prob) was requested with autograd"
            if why is None:
                return self._forward_impl(grad=True, **kw)
            if not why.startswith("the bound module has no trainable"):
                return self._grad_fallback(why, kw)
        return self._forward_nograd(**kw)

    def _forward_nograd(self, **kw) -> SDESchedulerOutput:
        with torch.no_grad():
            return self._forward_impl(grad=False, **kw)

    def _forward_impl(self, t, latents, prompt_embeds, pooled_prompt_embeds, img_ids, t_next, next_latents, guidance_scale, noise_level,
                      joint_attention_kwargs, compute_log_prob, return_kwargs, height, width, grad: bool) -> SDESchedulerOutput:
        self._before_engine_call()
        self._check_joint_attention_kwargs(joint_attention_kwargs)
        B, Ni, _ = latents.shape
        dev = latents.device
        # the latent grid is recovered from img_ids (rows / cols) when given, else it must be square or passed explicitly
        if img_ids is not None:
            hp, wp = int(img_ids[:, 1].max().item()) + 1, int(img_ids[:, 2].max().item()) + 1
        elif height is not None and width is not None:
            hp, wp = int(height) // 16, int(width) // 16
        else:
            hp = wp = int(round(Ni ** 0.5))
        if hp * wp != Ni:
            raise ValueError(f"mi355_flow: cannot recover the latent grid of {Ni} packed tokens (pass img_ids or height/width)")
        plan = self.engine.plan(B, 2 * hp, 2 * wp, prompt_embeds.shape[1], 1)
        t = torch.as_tensor(t, device=dev, dtype=torch.float32).reshape(-1)
        # timestep = t.expand(B) / 1000 (fp32); the model does `.to(latents.dtype) * 1000` (rounded again to that dtype)
        tm = ((t.double() / 1000).float().to(latents.dtype) * 1000).float()   # exact quotient, as the fused rollout's host math
        g = torch.as_tensor(guidance_scale, device=dev, dtype=latents.dtype).reshape(-1)
        gm = (g * 1000).float()
        gm = gm if self.engine.cfg.guidance_embeds else None
        if not grad:
            v = plan.transformer_forward(latents, tm, gm, prompt_embeds, pooled_prompt_embeds)
        sched = self.scheduler
        if t_next is None:
            idx = [sched.index_for_timestep(x) for x in t]
            t_next = torch.stack([sched.timesteps[j + 1] if j + 1 < len(sched.timesteps) else torch.zeros(()) for j in idx]).to(dev)
        t_next = torch.as_tensor(t_next, device=dev, dtype=torch.float32).reshape(-1)
        dyn = sched.dynamics_type
        sigma, sigma_next = (t.double() / 1000).float(), (t_next.double() / 1000).float()   # exact fp32 quotients (see adapter.forward)
        if sched.is_eval or dyn == "ODE":
            noise_level = 0.0
        elif noise_level is None:
            noise_level = sched.get_noise_level_for_sigma(sigma)
        noise = None
        if next_latents is None and dyn != "ODE":
            noise = randn_tensor(latents.shape, device=dev, dtype=torch.float32)
        view = (-1, 1, 1)
        if grad:
            from . import autograd as AG
            replay = next_latents is not None
            clp = bool(compute_log_prob) and replay
            call = dict(latents=latents, tm=tm, gm=gm, prompt_embeds=prompt_embeds, pooled=pooled_prompt_embeds, sigma=sigma, sigma_next=sigma_next,
                        eta=noise_level, sigma_max=float(sched.sigmas[1]), dynamics=dyn, next_latents=next_latents if replay else latents,
                        compute_log_prob=clp)
            lp, npred, mean, std, dtt = AG.flux_replay(self, plan, call)
            res = dict(noise_pred=npred, next_latents=next_latents.float() if replay else None, next_latents_mean=mean, std_dev_t=std.view(view),
                       dt=dtt.view(view), log_prob=lp if clp else None)
            return self._output_cls.from_dict({k: res[k] for k in return_kwargs if k in res})
        want = tuple(k for k in return_kwargs if k in ("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt"))
        o = sde_step(v, None, 1.0, latents, sigma, sigma_next, noise_level, float(sched.sigmas[1]), dyn, noise=noise,
                     next_latents=next_latents, compute_log_prob=compute_log_prob, want=want)
        res = dict(
            noise_pred=o.noise_pred,
            next_latents=o.next_latents if next_latents is None else next_latents.float(),
            next_latents_mean=o.next_latents_mean,
            std_dev_t=o.std_dev_t.view(view) if o.std_dev_t is not None else None,
            dt=o.dt.view(view) if o.dt is not None else None,
            log_prob=o.log_prob if compute_log_prob else None,
        )
        return self._output_cls.from_dict({k: res[k] for k in return_kwargs if k in res})


# ----------------------------------------------------------------------------- operator-level wrappers (tests, microbench)


class Flux1NativeAdapter(FluxRolloutMixin):
    """Standalone FLUX.1 adapter (no Flow-Factory import): engine + scheduler (+ optional native VAE decoder)."""

    def __init__(self, state_dict: Union[Dict[str, torch.Tensor], torch.nn.Module], config: Optional[FluxConfig] = None,
                 scheduler: Optional[FlowMatchEulerDiscreteSDEScheduler] = None, latent_storage_dtype: Optional[str] = "fp16",
                 transformer_dtype: torch.dtype = torch.bfloat16, device: Union[str, torch.device] = "cuda",
                 vae_state_dict: Optional[Dict[str, torch.Tensor]] = None, vae_config=None, vae_max_batch: int = 4):
        if not torch.cuda.is_available():
            raise RuntimeError("mi355_flow: no GPU visible; the native rollout engine has no CPU path")
        self.device = torch.device(device)
        self.transformer_dtype = transformer_dtype
        self._latent_storage = latent_storage_dtype
        # FLUX.1 scheduler config: dynamic shifting (mu from the image sequence length)
        self.scheduler = scheduler or FlowMatchEulerDiscreteSDEScheduler(shift=3.0, use_dynamic_shifting=True, sde_steps=[1, 2, 3],
                                                                         num_sde_steps=1)
        self.engine = FluxEngine(config or FluxConfig())
        self._live_weights = None
        if isinstance(state_dict, torch.nn.Module):
            # a torch module with HF parameter names (possibly DDP / peft wrapped): its CURRENT parameters are re-bound before every engine
            # call, and grad-mode forward() differentiates w.r.t. its trainable parameters (mi355_flow/autograd.py: flux_replay)
            from .binding import LiveWeights
            module = state_dict
            self._live_weights = LiveWeights(self.engine, lambda: module)
            self._sync_weights()
        else:
            self.refresh_weights(state_dict)
        self.vae_decoder = None
        self.vae_max_batch = vae_max_batch
        if vae_state_dict is not None:
            from .vae import VAEConfig, VAEDecoder
            self.vae_decoder = VAEDecoder(vae_config or VAEConfig(scaling_factor=0.3611, shift_factor=0.1159))
            self.vae_decoder.bind_state_dict(vae_state_dict)
            self.vae_decoder.ready()

    @property
    def latent_storage_dtype(self) -> Optional[torch.dtype]:
        return _DTYPE_MAP.get(self._latent_storage) if self._latent_storage else None

    def refresh_weights(self, state_dict: Dict[str, torch.Tensor]) -> None:
        self.engine.bind_state_dict(state_dict)
        self.engine.ready()

    def _sync_weights(self) -> int:
        if self._live_weights is None:
            return 0
        n = self._live_weights.sync()
        if n:
            self.engine.ready()
        return n

    def _before_engine_call(self) -> None:
        self._sync_weights()

    def rollout(self):
        self.scheduler.rollout()

    def eval(self):
        self.scheduler.eval()

    def train(self, mode: bool = True):
        self.scheduler.train(mode)

    def encode_prompt(self, *a, **k):
        raise RuntimeError("mi355_flow standalone adapter has no text encoders: pass prompt_embeds / pooled_prompt_embeds")

    def decode_latents(self, latents: torch.Tensor, height: int, width: int, output_type: str = "pt"):
        """flux1.py:138-147: unpack, / scaling + shift, vae.decode, postprocess."""
        if self.vae_decoder is None:
            return None
        if output_type not in ("pt", "np"):
            raise ValueError("mi355_flow standalone adapter decodes to 'pt' or 'np'")
        lat = unpack_latents(latents, int(height) // VAE_SCALE_FACTOR, int(width) // VAE_SCALE_FACTOR).contiguous()
        img = self.vae_decoder.decode(lat, postprocess=True, out_dtype=torch.bfloat16, max_batch=self.vae_max_batch)
        return img if output_type == "pt" else img.float().permute(0, 2, 3, 1).cpu().numpy()


def op_attention128(q: torch.Tensor, k: torch.Tensor, vT: torch.Tensor, S: int, n_first: Optional[int] = None, q_prescaled: bool = False):
    """q, k: [B, H, S_pad, 128] bf16; vT: [B, H, 128, S_pad] bf16 -> (o_first [B*n_first, H*128], o_rest [B*(S-n_first), H*128])."""
    lib = _lib.load()
    B, H, S_pad, _ = q.shape
    n_first = S if n_first is None else n_first
    D = H * 128
    o1 = torch.empty((B * n_first, D), device=q.device, dtype=torch.bfloat16)
    o2 = torch.empty((B * (S - n_first), D), device=q.device, dtype=torch.bfloat16) if n_first < S else None
    _lib.check(lib.mi355_op_attention128(_stream(), _ptr(q), _ptr(k), _ptr(vT), _ptr(o1) if n_first > 0 else None, D, n_first, _ptr(o2), D, B, H, S, S_pad,
                                         int(q_prescaled)), "op_attention128")
    return o1, o2


def op_rope_norm(src: torch.Tensor, q_col: int, k_col: int, nw_q: torch.Tensor, nw_k: torch.Tensor, cos_sin: torch.Tensor, B: int, H: int,
                 rows_per_sample: int, s_off: int, S_pad: int, eps: float = 1e-6, q_scale: float = 1.0):
    """src [M, ld] bf16 (q at q_col, k at k_col); cos_sin [S, 64, 2] fp32 -> q, k [B, H, S_pad, 128] bf16 (zero elsewhere)."""
    lib = _lib.load()
    M = src.shape[0]
    q = torch.zeros((B, H, S_pad, 128), device=src.device, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    _lib.check(lib.mi355_op_rope_norm(_stream(), _ptr(src), src.shape[1], q_col, k_col, _ptr(nw_q), _ptr(nw_k), _ptr(cos_sin), _ptr(q), _ptr(k),
                                      M, H, rows_per_sample, s_off, S_pad, eps, q_scale), "op_rope_norm")
    return q, k
