"""Wan2.1 text-to-video rollout on the native engine (mi355_wan_*): host mirror of `Wan2_T2V_Adapter.inference` / `.forward`
(reference src/flow_factory/models/wan/wan2_t2v.py:234-421, :426-543) -- SURVEY.md 8(f) row N4, single-transformer Wan2.1.

Kept from the reference: latents `(B, 16, T, h, w)` drawn in fp32 (`prepare_latents(dtype=float32)`), `timestep = t.expand(B)` with
the scheduler's integer timesteps, classifier-free guidance as `u + g (c - u)` (here one forward over the batch [negative, positive]
instead of two passes), `UniPCMultistepSDEScheduler.step` in rollout mode = the four SDE / ODE dynamics with sigma = t / 1000.
Not covered (raise): `attention_kwargs`; image-conditioned TI2V (per-token timesteps that DIFFER: the text-to-video use of Wan2.2-TI2V-5B, whose
`expand_timesteps` mask is all ones, runs as the scalar-timestep forward).  The causal 3-D video VAE decode is native too
(`mi355_flow.vae.WanVAEDecoder`, csrc/wan_vae_engine.hip): pass `vae_state_dict` to the standalone adapter (or a `video_decode` callable).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import DYNAMICS, WanCfg
from .engine import WeightHolder, _bf16c, _ptr, _stream, dtype_code, sde_step
from .samples import WanT2VSample
from .scheduler import FlowMatchEulerDiscreteSDEScheduler, SDESchedulerOutput, host_noise_levels, randn_tensor
from .trajectory import TrajectoryIndicesType, _resolve, collect_rollout

_DTYPE_MAP = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16,
              "fp32": torch.float32, "float32": torch.float32}
VAE_SCALE_SPATIAL, VAE_SCALE_TEMPORAL = 8, 4


@dataclass
class WanConfig:
    """diffusers WanTransformer3DModel config fields the engine needs (Wan2.1-T2V-1.3B defaults)."""
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

    def to_c(self) -> WanCfg:
        return WanCfg(self.in_channels, self.out_channels, self.num_layers, self.num_attention_heads, self.attention_head_dim, self.ffn_dim,
                      self.text_dim, self.freq_dim, self.patch_size[0], self.patch_size[1], self.patch_size[2], self.eps)


class UniPCMultistepSDEScheduler(FlowMatchEulerDiscreteSDEScheduler):
    """Rollout-mode mirror of the reference's `UniPCMultistepSDEScheduler` (scheduler/unipc_multistep.py): the SDE-step selection
    mixin and `step()` are shared with the flow-match scheduler (its train / rollout branch is the same four dynamics,
    :296-421); the schedule is diffusers' UniPC flow schedule (`use_flow_sigmas`, `flow_shift`): integer timesteps."""

    def __init__(self, flow_shift: float = 3.0, **kw):
        kw.setdefault("shift", flow_shift)
        super().__init__(**kw)
        self.config["flow_shift"] = flow_shift

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, **_ignored) -> None:
        n_train = self.config.num_train_timesteps
        shift = self.config["flow_shift"]
        alphas = np.linspace(1, 1 / n_train, num_inference_steps + 1)
        sig = 1.0 - alphas
        sig = np.flip(shift * sig / (1 + (shift - 1) * sig))[:-1].copy()
        ts = (sig * n_train).copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device=device)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32)).to(device=device)
        self.num_inference_steps = len(ts)
        self._host_timesteps = [float(x) for x in ts.tolist()]


class WanEngine(WeightHolder):
    def __init__(self, cfg: WanConfig = WanConfig()):
        self.lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        c = cfg.to_c()
        _lib.check(self.lib.mi355_wan_create(C.byref(c), C.byref(h)), "wan_create")
        self._h = h
        self._plans: Dict[tuple, "WanPlan"] = {}

    _ABI, _WHAT = "wan", "Wan transformer"

    def plan(self, batch: int, n_cfg: int, T: int, h: int, w: int, n_text: int, max_steps: int) -> "WanPlan":
        key = (batch, n_cfg, T, h, w, n_text)
        p = self._plans.get(key)
        if p is None or p.max_steps < max_steps:
            if p is not None:
                p.close()
            p = WanPlan(self, batch, n_cfg, T, h, w, n_text, max_steps)
            self._plans[key] = p
        return p

    # ---------------------------------------------------------------- weight gradients (mi355_flow/autograd.py: wan_replay)
    #: the native backward (end of round 4; its seven GPU tests are green on MI355X: profiles/r04u_*, r04v_*).  MI355_WAN_NATIVE_BACKWARD=0 keeps
    #: grad-mode forward() on the engine-valued replay (value = engine, gradient = the reference's torch path)
    native_backward_enabled = os.environ.get("MI355_WAN_NATIVE_BACKWARD", "1") != "0"

    def grad_supported(self, name: str) -> int:
        """1 = the native backward produces a gradient for this parameter (the linear layers inside the transformer blocks), 0 = it does not."""
        if not type(self).native_backward_enabled:
            return 0
        return 1 if self.lib.mi355_wan_grad_supported(self._h, name.encode()) == 0 else 0

    def set_grad(self, name: str, grad: torch.Tensor) -> None:
        if grad.dtype not in (torch.float32, torch.bfloat16) or not grad.is_contiguous():
            raise ValueError("mi355_flow: gradient buffers are contiguous fp32 (or bf16) tensors")
        _lib.check(self.lib.mi355_wan_set_grad_typed(self._h, name.encode(), _ptr(grad), dtype_code(grad.dtype)), f"wan_set_grad({name})")

    def clear_grads(self) -> None:
        _lib.check(self.lib.mi355_wan_clear_grads(self._h), "wan_clear_grads")

    def close(self) -> None:
        for p in self._plans.values():
            p.close()
        self._plans.clear()
        if self._h:
            self.lib.mi355_wan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WanPlan:
    def __init__(self, engine: WanEngine, batch: int, n_cfg: int, T: int, h: int, w: int, n_text: int, max_steps: int):
        self.engine, self.lib = engine, engine.lib
        self.batch, self.n_cfg, self.T, self.h, self.w, self.n_text, self.max_steps = batch, n_cfg, T, h, w, n_text, max_steps
        self.C = engine.cfg.in_channels
        hd = C.c_void_p()
        _lib.check(self.lib.mi355_wan_plan_create(engine._h, batch, n_cfg, T, h, w, n_text, max_steps, C.byref(hd)), "wan_plan_create")
        self._h = hd

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.mi355_wan_plan_workspace_bytes(self._h))

    def close(self) -> None:
        if self._h:
            self.lib.mi355_wan_plan_destroy(self._h)
            self._h = None

    def transformer_forward(self, latents: torch.Tensor, t: torch.Tensor, enc_a: torch.Tensor, enc_b: Optional[torch.Tensor] = None):
        """latents (B, 16, T, h, w); t (B*n_cfg,) or scalar; n_cfg == 2: enc_a = negative, enc_b = positive, output (2B, ...)."""
        B, Bp = self.batch, self.batch * self.n_cfg
        assert tuple(latents.shape) == (B, self.C, self.T, self.h, self.w), latents.shape
        dev = latents.device
        t = t.to(device=dev, dtype=torch.float32).reshape(-1)
        t = (t.expand(Bp) if t.numel() == 1 else (t.repeat(self.n_cfg) if t.numel() == B and self.n_cfg == 2 else t)).contiguous()
        assert t.numel() == Bp
        out = torch.empty((Bp, self.engine.cfg.out_channels, self.T, self.h, self.w), device=dev, dtype=torch.bfloat16)
        latents = latents.contiguous()
        ea = _bf16c(enc_a)
        eb = _bf16c(enc_b) if enc_b is not None else None
        _lib.check(self.lib.mi355_wan_forward(self._h, _stream(), _ptr(latents), dtype_code(latents.dtype), _ptr(t), _ptr(ea), _ptr(eb),
                                              _ptr(out)), "wan_forward")
        return out

    # ---------------------------------------------------------------- differentiable forward (optimize() replay)
    def forward_train(self, latents: torch.Tensor, t: torch.Tensor, enc_a: torch.Tensor, enc_b: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mi355_wan_forward_train: `transformer_forward` on per-block activation buffers -- the same kernel binaries, so the prediction is
        bit-identical -- keeping what `backward` needs in the plan's training stash (ONE per plan; every call takes a serial number)."""
        B, Bp = self.batch, self.batch * self.n_cfg
        assert tuple(latents.shape) == (B, self.C, self.T, self.h, self.w), latents.shape
        dev = latents.device
        t = t.to(device=dev, dtype=torch.float32).reshape(-1)
        t = (t.expand(Bp) if t.numel() == 1 else (t.repeat(self.n_cfg) if t.numel() == B and self.n_cfg == 2 else t)).contiguous()
        assert t.numel() == Bp
        out = torch.empty((Bp, self.engine.cfg.out_channels, self.T, self.h, self.w), device=dev, dtype=torch.bfloat16)
        latents = latents.contiguous()
        ea = _bf16c(enc_a)
        eb = _bf16c(enc_b) if enc_b is not None else None
        _lib.check(self.lib.mi355_wan_forward_train(self._h, _stream(), _ptr(latents), dtype_code(latents.dtype), _ptr(t), _ptr(ea), _ptr(eb),
                                                    _ptr(out)), "wan_forward_train")
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        return out

    def backward(self, dv: torch.Tensor) -> None:
        """mi355_wan_backward: d loss / d v [n_cfg * B, C, T, h, w] fp32 ([uncond | text], as `engine.sde_step_bwd` returns it) of the LAST
        `forward_train` -> the buffers registered with `WanEngine.set_grad`."""
        dv = dv.to(torch.float32).contiguous()
        assert tuple(dv.shape) == (self.batch * self.n_cfg, self.engine.cfg.out_channels, self.T, self.h, self.w), dv.shape
        _lib.check(self.lib.mi355_wan_backward(self._h, _stream(), _ptr(dv)), "wan_backward")

    @property
    def training_bytes(self) -> int:
        return int(self.lib.mi355_wan_plan_training_bytes(self._h))

    def rollout(self, timesteps: Sequence[float], sigmas: Sequence[float], noise_levels: Sequence[float], dynamics: str, guidance: float,
                init_latents: torch.Tensor, storage_dtype: torch.dtype, step_noise: Optional[torch.Tensor], prompt_embeds: torch.Tensor,
                neg_embeds: Optional[torch.Tensor] = None, keep_positions: Optional[Sequence[int]] = None, compute_log_prob: bool = True):
        N, B = len(timesteps), self.batch
        assert len(sigmas) == N + 1 and len(noise_levels) == N
        dev = init_latents.device
        keep = list(range(N + 1)) if keep_positions is None else sorted(set(int(k) for k in keep_positions))
        slots = [-1] * (N + 1)
        for s, pos in enumerate(keep):
            slots[pos] = s
        shape = (B, self.C, self.T, self.h, self.w)
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
            assert step_noise.dtype == torch.float32 and tuple(step_noise.shape) == (N,) + shape, step_noise.shape
        pe = _bf16c(prompt_embeds)
        ne = _bf16c(neg_embeds) if neg_embeds is not None else None
        _lib.check(self.lib.mi355_wan_rollout(
            self._h, _stream(), N, ts_c, sg_c, nl_c, DYNAMICS[dynamics], float(guidance), _ptr(init_latents), dtype_code(init_latents.dtype),
            dtype_code(storage_dtype), _ptr(step_noise), _ptr(pe), _ptr(ne), sl_c, _ptr(out_lat), _ptr(out_lp), _ptr(out_fin),
            int(bool(compute_log_prob))), "wan_rollout")
        return out_lat, out_lp, out_fin


class WanRolloutMixin:
    """`inference()` / `forward()` of `Wan2_T2V_Adapter` (reference models/wan/wan2_t2v.py:234-421, :426-543) on the engine, for the
    single-transformer Wan2.1 configuration.  Host classes provide `engine` (WanEngine), `scheduler`, `device`, `transformer_dtype`,
    `latent_storage_dtype`, `encode_prompt`, `decode_latents(latents, output_type)`.

    Evaluation mode: the reference's `UniPCMultistepSDEScheduler.step` delegates to diffusers' UniPC multistep predictor-corrector when
    `is_eval` (scheduler/unipc_multistep.py:282-285).  Since round 5 `inference()` samples it natively (`_rollout_eval`: per-step engine
    forwards + the solver as two streaming HIP kernels, mi355_flow/unipc.py; the solver body is third-party code restated from its published
    algorithm -- oracle/unipc_ref.py, parity unpinned).  `MI355_WAN_EVAL_REFERENCE=1` routes evaluation to `_eval_inference` instead (the
    Flow-Factory plugin: the reference's own loop; standalone it raises) -- never a silent first-order Euler step."""

    _sample_cls = WanT2VSample
    _output_cls = SDESchedulerOutput
    # Wan2.2 two-expert pipelines (wan2_t2v.py:476-487): `engine` holds the high-noise expert (`pipeline.transformer`, used while
    # t >= boundary_ratio * num_train_timesteps with `guidance_scale`), `engine_2` the low-noise one (`pipeline.transformer_2`, with
    # `guidance_scale_2`).  None = single-transformer Wan2.1.  Per-token timesteps (`expand_timesteps`, TI2V-5B) are not supported.
    engine_2 = None
    boundary_ratio: Optional[float] = None
    low_noise_only = False      # a Wan2.2 pipeline whose boundary lies above every timestep: `engine` holds transformer_2, guidance_scale_2 applies

    def _boundary_timestep(self, boundary_timestep: Optional[float] = None) -> Optional[float]:
        if boundary_timestep is None and self.boundary_ratio is not None:
            boundary_timestep = float(self.boundary_ratio) * float(self.scheduler.config.get("num_train_timesteps", 1000))
        return boundary_timestep

    def _expert(self, t: float, guidance_scale: float, guidance_scale_2: Optional[float], boundary_timestep: Optional[float]):
        """(engine, guidance) for a step at timestep t: the reference's selection rule."""
        if self.low_noise_only:
            return self.engine, (guidance_scale_2 if guidance_scale_2 is not None else guidance_scale)
        bt = self._boundary_timestep(boundary_timestep)
        if self.engine_2 is None or bt is None or t >= bt:
            return self.engine, guidance_scale
        return self.engine_2, (guidance_scale_2 if guidance_scale_2 is not None else guidance_scale)

    def _before_engine_call(self) -> None:
        """Hook run at the top of inference() / forward(): the Flow-Factory plugin re-binds changed weights here."""

    def _eval_inference(self, **kwargs):
        raise NotImplementedError("mi355_flow: evaluation-mode sampling for Wan uses diffusers' UniPC multistep solver in the reference "
                                  "(unipc_multistep.py:282-285); the native engine implements the rollout (SDE / Euler) branch only")

    def cast_latents(self, latents: torch.Tensor, default_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        target = self.latent_storage_dtype or default_dtype
        if target is None or latents.dtype == target:
            return latents
        if target == torch.float16:
            latents = latents.clamp(-65504.0, 65504.0)
        return latents.to(target)

    # ------------------------------------------------------------------ rollout (wan2_t2v.py:234-421)
    @torch.no_grad()
    def inference(
        self,
        prompt: Optional[Union[str, List[str]]] = None,
        negative_prompt: Optional[Union[str, List[str]]] = None,
        height: int = 480,
        width: int = 832,
        num_frames: int = 81,
        num_inference_steps: int = 50,
        guidance_scale: float = 5.0,
        guidance_scale_2: Optional[float] = None,
        generator: Optional[torch.Generator] = None,
        prompt_ids: Optional[torch.Tensor] = None,
        prompt_embeds: Optional[torch.Tensor] = None,
        negative_prompt_ids: Optional[torch.Tensor] = None,
        negative_prompt_embeds: Optional[torch.Tensor] = None,
        compute_log_prob: bool = False,
        attention_kwargs: Optional[Dict[str, Any]] = None,
        max_sequence_length: int = 512,
        extra_call_back_kwargs: List[str] = [],
        trajectory_indices: TrajectoryIndicesType = "all",
    ) -> List[WanT2VSample]:
        device = self.device
        eval_mode = bool(getattr(self.scheduler, "is_eval", False))
        if eval_mode and os.environ.get("MI355_WAN_EVAL_REFERENCE") == "1":
            return self._eval_inference(
                prompt=prompt, negative_prompt=negative_prompt, height=height, width=width, num_frames=num_frames,
                num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, guidance_scale_2=guidance_scale_2,
                generator=generator, prompt_ids=prompt_ids, prompt_embeds=prompt_embeds, negative_prompt_ids=negative_prompt_ids,
                negative_prompt_embeds=negative_prompt_embeds, compute_log_prob=compute_log_prob, attention_kwargs=attention_kwargs,
                max_sequence_length=max_sequence_length, extra_call_back_kwargs=extra_call_back_kwargs,
                trajectory_indices=trajectory_indices)
        self._before_engine_call()
        if attention_kwargs:
            raise NotImplementedError("mi355_flow: attention_kwargs are not supported by the native engine")
        if self.low_noise_only:
            if guidance_scale_2 is not None:
                guidance_scale = guidance_scale_2          # every step runs the low-noise expert with its own scale
            guidance_scale_2 = None
        if guidance_scale_2 is not None and guidance_scale_2 != guidance_scale and self.engine_2 is None:
            raise ValueError("mi355_flow: guidance_scale_2 needs a two-expert (Wan2.2) adapter: no second transformer is bound")
        # (the pipeline's VAE compression: 4 x 8 x 8 for Wan2.1 / Wan2.2-A14B, 4 x 16 x 16 for the Wan2.2-TI2V-5B VAE; the plugin reads them
        # from the pipeline, wan2_t2v.py:267-279)
        vst, vss = int(getattr(self, "vae_scale_temporal", VAE_SCALE_TEMPORAL)), int(getattr(self, "vae_scale_spatial", VAE_SCALE_SPATIAL))
        if (num_frames - 1) % vst != 0:
            num_frames = num_frames // vst * vst + 1
        num_frames = max(num_frames, 1)
        ps = self.engine.cfg.patch_size
        hm, wm = vss * ps[1], vss * ps[2]
        height, width = height // hm * hm, width // wm * wm
        if prompt_embeds is None:
            enc = self.encode_prompt(prompt=prompt, negative_prompt=negative_prompt, guidance_scale=guidance_scale)
            prompt_embeds, prompt_ids = enc["prompt_embeds"], enc["prompt_ids"]
            negative_prompt_embeds, negative_prompt_ids = enc.get("negative_prompt_embeds"), enc.get("negative_prompt_ids")
        prompt_embeds = prompt_embeds.to(device).to(self.transformer_dtype)
        if negative_prompt_embeds is not None:
            negative_prompt_embeds = negative_prompt_embeds.to(device).to(self.transformer_dtype)
        two = self.engine_2 is not None
        g2 = guidance_scale_2 if guidance_scale_2 is not None else guidance_scale
        do_cfg = (guidance_scale > 1.0 or (two and g2 > 1.0)) and negative_prompt_embeds is not None
        B = prompt_embeds.shape[0]
        N = int(num_inference_steps)
        self.scheduler.set_timesteps(N, device=device)
        timesteps = self.scheduler.timesteps
        Cl = self.engine.cfg.in_channels
        T, h, w = (num_frames - 1) // vst + 1, height // vss, width // vss
        # RNG in the reference's order: prepare_latents in fp32, then one fp32 draw per step
        # (none of the step draws under ODE dynamics: the reference's ODE branch draws nothing)
        latents = randn_tensor((B, Cl, T, h, w), generator=generator, device=device, dtype=torch.float32)
        step_noise = None
        if self.scheduler.dynamics_type != "ODE" and not eval_mode:      # (the evaluation-mode solver is deterministic: the reference draws nothing)
            step_noise = torch.empty((N, B, Cl, T, h, w), device=device, dtype=torch.float32)
            for i in range(N):
                step_noise[i] = randn_tensor((B, Cl, T, h, w), generator=None, device=device, dtype=torch.float32)
        ts_host = [float(t) for t in timesteps.tolist()]
        sig_host = [float(s) for s in self.scheduler.sigmas.tolist()]
        eta_host = host_noise_levels(self.scheduler, N)
        storage = self.latent_storage_dtype or torch.float32       # cast_latents(latents) with no default: fp32 stays fp32
        # two experts: per-step engine calls (each step picks its expert and guidance by the boundary rule; still all-HIP)
        stepwise = two or any(k != "noise_level" for k in extra_call_back_kwargs)
        plan = self.engine.plan(B, 2 if do_cfg else 1, T, h, w, prompt_embeds.shape[1], N)
        plan_2 = self.engine_2.plan(B, 2 if do_cfg else 1, T, h, w, prompt_embeds.shape[1], N) if two else None
        kept = _resolve(trajectory_indices, N + 1)
        keep_positions = list(range(N + 1)) if kept is None else sorted(kept)
        step_outputs = None
        if eval_mode:
            if compute_log_prob or any(k != "noise_level" for k in extra_call_back_kwargs):
                raise NotImplementedError("mi355_flow: evaluation-mode Wan sampling returns latents only (the reference's UniPC step yields neither "
                                          "a log-prob nor per-step callback tensors: unipc_multistep.py:282-285)")
            lat_kept = self._rollout_eval(plan, ts_host, sig_host, guidance_scale, latents, storage, prompt_embeds,
                                          negative_prompt_embeds if do_cfg else None, plan_2=plan_2, guidance_2=g2)
            log_probs = torch.full((N, B), float("nan"), device=device)
            final = lat_kept[N]
            pos_to_slot = {p: p for p in range(N + 1)}
        elif not stepwise:
            lat_kept, log_probs, final = plan.rollout(ts_host, sig_host, eta_host, self.scheduler.dynamics_type, guidance_scale, latents,
                                                      storage, step_noise, prompt_embeds, negative_prompt_embeds if do_cfg else None,
                                                      keep_positions=keep_positions, compute_log_prob=compute_log_prob)
            pos_to_slot = {p: s for s, p in enumerate(keep_positions)}
        else:
            lat_kept, log_probs, step_outputs = self._rollout_stepwise(plan, ts_host, sig_host, eta_host, guidance_scale, latents, storage,
                                                                       step_noise, prompt_embeds, negative_prompt_embeds if do_cfg else None,
                                                                       compute_log_prob, extra_call_back_kwargs, plan_2=plan_2,
                                                                       guidance_2=g2)
            final = lat_kept[N]
            pos_to_slot = {p: p for p in range(N + 1)}
        traj = collect_rollout(trajectory_indices, N, lambda pos: lat_kept[pos_to_slot[pos]], log_probs, eta_host, compute_log_prob,
                               step_outputs, extra_call_back_kwargs,
                               captured_noise_levels=host_noise_levels(self.scheduler, N, effective=False), dynamics=self.scheduler.dynamics_type)
        videos = self.decode_latents(final, output_type="pt")
        return [
            self._sample_cls(
                timesteps=timesteps,
                **traj.per_sample(b),
                video=videos[b] if videos is not None else None,
                height=height, width=width,
                prompt=prompt[b] if isinstance(prompt, list) else prompt,
                prompt_ids=prompt_ids[b] if prompt_ids is not None else None,
                prompt_embeds=prompt_embeds[b],
                negative_prompt=negative_prompt[b] if isinstance(negative_prompt, list) else negative_prompt,
                negative_prompt_ids=negative_prompt_ids[b] if negative_prompt_ids is not None else None,
                negative_prompt_embeds=negative_prompt_embeds[b] if negative_prompt_embeds is not None else None,
            )
            for b in range(B)
        ]

    def _rollout_eval(self, plan, ts, sig, guidance, latents, storage, pe, ne, plan_2=None, guidance_2=None):
        """Evaluation-mode sampling (reference: the loop of wan2_t2v.py:346-375 with `scheduler.step` in its `is_eval` branch,
        unipc_multistep.py:282-285 = diffusers' UniPC multistep predictor-corrector): one engine forward per step (per expert and guidance by
        the boundary rule, like the rollout), then the solver step on the GPU -- `mi355_unipc_convert` + `mi355_op_lincomb` with host-side
        coefficients (mi355_flow/unipc.py).  Deterministic: no noise is drawn."""
        from .unipc import UniPCSampler
        cfg = self.scheduler.config
        get = (lambda k, d: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d: getattr(cfg, k, d))
        if get("prediction_type", "flow_prediction") != "flow_prediction" or not get("predict_x0", True):
            raise NotImplementedError("mi355_flow: the native UniPC sampler implements flow prediction with predict_x0 (the Wan pipelines' setting)")
        if get("thresholding", False) or get("solver_p", None) is not None:
            raise NotImplementedError("mi355_flow: UniPC thresholding / a custom predictor solver are not implemented")
        solver = UniPCSampler(sig, solver_order=int(get("solver_order", 2)), solver_type=str(get("solver_type", "bh2")),
                              lower_order_final=bool(get("lower_order_final", True)), disable_corrector=tuple(get("disable_corrector", ()) or ()),
                              sample_dtype=self.cast_latents(latents[:0], storage).dtype)
        N, B = len(ts), latents.shape[0]
        cur = self.cast_latents(latents, storage)
        all_lat = [cur]
        f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)          # noqa: E731
        bt = self._boundary_timestep()
        for i in range(N):
            low = plan_2 is not None and bt is not None and ts[i] < bt
            pl, gd = (plan_2, guidance_2) if low else (plan, guidance)
            cfg_i = ne is not None and gd > 1.0
            if not cfg_i and pl.n_cfg == 2:
                pl = pl.engine.plan(B, 1, pl.T, pl.h, pl.w, pl.n_text, 1)
            v = pl.transformer_forward(cur, f32(ts[i]).reshape(1), ne if cfg_i else pe, pe if cfg_i else None)
            vu, vt = (v[:B], v[B:]) if cfg_i else (None, v)
            nxt = solver.step(i, vt.reshape(cur.shape), vu.reshape(cur.shape) if vu is not None else None, gd, cur)
            cur = self.cast_latents(nxt, storage)
            all_lat.append(cur)
        return all_lat

    def _rollout_stepwise(self, plan, ts, sig, eta, guidance, latents, storage, step_noise, pe, ne, compute_log_prob, extra_keys, plan_2=None,
                          guidance_2=None):
        """Per-step engine calls (still all-HIP) for rollouts that ask for per-step callback tensors, and for two-expert pipelines."""
        N, B = len(ts), latents.shape[0]
        cur = self.cast_latents(latents, storage)
        all_lat = [cur]
        log_probs = torch.full((N, B), float("nan"), device=latents.device)
        outs = []
        want = tuple(k for k in extra_keys if k in ("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt"))
        f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
        for i in range(N):
            t_next = ts[i + 1] if i + 1 < N else 0.0
            clp = compute_log_prob and eta[i] > 0
            bt = self._boundary_timestep()
            low = plan_2 is not None and bt is not None and ts[i] < bt
            pl, gd = (plan_2, guidance_2) if low else (plan, guidance)
            cfg_i = ne is not None and gd > 1.0          # the reference decides CFG per expert (wan2_t2v.py:489-498)
            if not cfg_i and pl.n_cfg == 2:
                pl = pl.engine.plan(B, 1, pl.T, pl.h, pl.w, pl.n_text, 1)
            v = pl.transformer_forward(cur, f32(ts[i]).reshape(1), ne if cfg_i else pe, pe if cfg_i else None)
            vu, vt = (v[:B], v[B:]) if cfg_i else (None, v)
            o = sde_step(vt, vu, gd, cur, f32(ts[i]) / f32(1000.0), f32(t_next) / f32(1000.0), eta[i], sig[1],
                         self.scheduler.dynamics_type, noise=step_noise[i] if step_noise is not None else None, compute_log_prob=clp,
                         want=want)
            if clp:
                log_probs[i] = o.log_prob
            cur = o.next_storage
            all_lat.append(cur)
            outs.append(o)
        return all_lat, log_probs, outs

    # ------------------------------------------------------------------ single step / replay (wan2_t2v.py:426-543), no-grad
    def forward(
        self,
        t: torch.Tensor,
        latents: torch.Tensor,
        prompt_embeds: torch.Tensor,
        negative_prompt_embeds: Optional[torch.Tensor] = None,
        guidance_scale: float = 5.0,
        guidance_scale_2: Optional[float] = None,
        t_next: Optional[torch.Tensor] = None,
        next_latents: Optional[torch.Tensor] = None,
        noise_level: Optional[float] = None,
        attention_kwargs: Optional[Dict[str, Any]] = None,
        compute_log_prob: bool = True,
        return_kwargs: List[str] = ["noise_pred", "next_latents", "next_latents_mean", "std_dev_t", "dt", "log_prob"],
        boundary_timestep: Optional[float] = None,
    ) -> SDESchedulerOutput:
        kw = dict(t=t, latents=latents, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, guidance_scale=guidance_scale,
                  guidance_scale_2=guidance_scale_2, t_next=t_next, next_latents=next_latents, noise_level=noise_level,
                  attention_kwargs=attention_kwargs, compute_log_prob=compute_log_prob, return_kwargs=return_kwargs, boundary_timestep=boundary_timestep)
        if torch.is_grad_enabled() and getattr(self, "_live_weights", None) is not None and WanEngine.native_backward_enabled:
            # optimize() (trainers/grpo.py:263): the replay WITH autograd on the engine's differentiable forward + native backward
            # (mi355_flow.autograd.wan_replay) when its backward covers the trainable set of the transformer THIS step runs on (Wan2.2: the
            # expert the timestep selects, wan2_t2v.py:476-487).  MI355_WAN_NATIVE_BACKWARD=0 opts out.
            from . import autograd as AG
            self._before_engine_call()
            t0 = float(torch.as_tensor(t, dtype=torch.float32).reshape(-1)[0])
            host = self._replay_host(self._expert(t0, guidance_scale, guidance_scale_2, boundary_timestep)[0])
            why = AG.unsupported_reason(host)
            sampled = next_latents is None and ("next_latents" in return_kwargs or (compute_log_prob and "log_prob" in return_kwargs))
            if why is None and sampled:
                why = "a sampled next state (or its log-prob) was requested with autograd"
            if why is None:
                return self._forward_impl(grad=True, host=host, **kw)
            if not why.startswith("the bound module has no trainable"):
                return self._grad_fallback(why, kw)
        return self._forward_nograd(**kw)

    def _forward_nograd(self, **kw) -> SDESchedulerOutput:
        with torch.no_grad():
            return self._forward_impl(grad=False, **kw)

    def _grad_fallback(self, why: str, kwargs: Dict[str, Any]):
        """Grad-mode forward() the native backward cannot serve.  Standalone: there is no other implementation -- raise (the Flow-Factory
        plugin overrides this with the reference's autograd path)."""
        raise NotImplementedError(f"mi355_flow: Wan forward() with autograd is not available natively: {why}")

    def _replay_host(self, eng):
        """What `mi355_flow.autograd` differentiates against for a step on `eng`: this adapter (single transformer / the high-noise expert), or
        the low-noise expert's (engine, live module binding) pair of a two-expert Wan2.2 pipeline."""
        if eng is self.engine or self.engine_2 is None:
            return self
        import types
        return types.SimpleNamespace(engine=self.engine_2, _live_weights=getattr(self, "_live_weights_2", None),
                                     _sync_weights=getattr(self, "_sync_weights", None))

    def _forward_impl(self, t, latents, prompt_embeds, negative_prompt_embeds, guidance_scale, guidance_scale_2, t_next, next_latents, noise_level,
                      attention_kwargs, compute_log_prob, return_kwargs, boundary_timestep, grad: bool, host=None) -> SDESchedulerOutput:
        self._before_engine_call()
        if attention_kwargs:
            raise NotImplementedError("mi355_flow: attention_kwargs are not supported by the native engine")
        if boundary_timestep is not None and self.engine_2 is None:
            raise ValueError("mi355_flow: boundary_timestep needs a two-expert (Wan2.2) adapter: no second transformer is bound")
        dev = latents.device
        B, _, T, h, w = latents.shape
        t = torch.as_tensor(t, device=dev, dtype=torch.float32).reshape(-1)
        t0 = t[0]                                       # `t = t[0] if t.ndim == 1 else t`: one scalar timestep per call
        sched = self.scheduler
        if t_next is None:
            idx = sched.index_for_timestep(t0)
            t_next = sched.timesteps[idx + 1].float() if idx + 1 < len(sched.timesteps) else torch.zeros(())
        t_next = torch.as_tensor(t_next, device=dev, dtype=torch.float32).reshape(-1)[0]
        eng, guidance_scale = self._expert(float(t0), guidance_scale, guidance_scale_2, boundary_timestep)
        do_cfg = negative_prompt_embeds is not None and guidance_scale > 1.0
        plan = eng.plan(B, 2 if do_cfg else 1, T, h, w, prompt_embeds.shape[1], 1)
        enc_a, enc_b = (negative_prompt_embeds, prompt_embeds) if do_cfg else (prompt_embeds, None)
        if not grad:
            v = plan.transformer_forward(latents, t0.reshape(1), enc_a, enc_b)
            vu, vt = (v[:B], v[B:]) if do_cfg else (None, v)
        dyn = sched.dynamics_type
        sigma, sigma_next = (t0.double() / 1000).float(), (t_next.double() / 1000).float()
        if sched.is_eval or dyn == "ODE":
            noise_level = 0.0
        elif noise_level is None:
            noise_level = sched.get_noise_level_for_timestep(float(t0))
        noise = None
        if next_latents is None and dyn != "ODE":
            noise = randn_tensor(latents.shape, device=dev, dtype=torch.float32)
        view = (-1, 1, 1, 1, 1)
        if grad:
            from . import autograd as AG
            replay = next_latents is not None
            clp = bool(compute_log_prob) and replay
            call = dict(latents=latents, train_args=(latents, t0.reshape(1), enc_a, enc_b), cfg_guidance=float(guidance_scale) if do_cfg else None,
                        sigma=sigma, sigma_next=sigma_next, eta=noise_level, sigma_max=float(sched.sigmas[1]), dynamics=dyn,
                        next_latents=next_latents if replay else latents, compute_log_prob=clp)
            lp, npred, mean, std, dtt = AG.wan_replay(host if host is not None else self, plan, call)
            res = dict(noise_pred=npred, next_latents=next_latents.float() if replay else None, next_latents_mean=mean, std_dev_t=std.view(view),
                       dt=dtt.view(view), log_prob=lp if clp else None)
            return self._output_cls.from_dict({k: res[k] for k in return_kwargs if k in res})
        want = tuple(k for k in return_kwargs if k in ("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt"))
        o = sde_step(vt, vu, guidance_scale, latents, sigma, sigma_next, noise_level, float(sched.sigmas[1]), dyn, noise=noise,
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


class Wan2T2VNativeAdapter(WanRolloutMixin):
    """Standalone Wan2.1 T2V adapter (no Flow-Factory import): engine + UniPC-SDE scheduler (+ optional video decoder callable)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[WanConfig] = None,
                 scheduler: Optional[UniPCMultistepSDEScheduler] = None, latent_storage_dtype: Optional[str] = "fp16",
                 transformer_dtype: torch.dtype = torch.bfloat16, device: Union[str, torch.device] = "cuda",
                 video_decode: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 state_dict_2: Optional[Dict[str, torch.Tensor]] = None, boundary_ratio: Optional[float] = None,
                 vae_state_dict: Optional[Dict[str, torch.Tensor]] = None, vae_config=None, vae_max_batch: int = 1):
        if not torch.cuda.is_available():
            raise RuntimeError("mi355_flow: no GPU visible; the native rollout engine has no CPU path")
        self.device = torch.device(device)
        self.transformer_dtype = transformer_dtype
        self._latent_storage = latent_storage_dtype
        self.scheduler = scheduler or UniPCMultistepSDEScheduler(flow_shift=3.0, sde_steps=[1, 2, 3], num_sde_steps=1)
        self.engine = WanEngine(config or WanConfig())
        self._live_weights = None
        if isinstance(state_dict, torch.nn.Module):
            # a torch module with HF parameter names (possibly DDP / peft wrapped): its CURRENT parameters are re-bound before every engine
            # call, and grad-mode forward() differentiates w.r.t. its trainable parameters (mi355_flow/autograd.py: wan_replay)
            from .binding import LiveWeights
            module = state_dict
            self._live_weights = LiveWeights(self.engine, lambda: module)
            self._sync_weights()
        else:
            self.refresh_weights(state_dict)
        if state_dict_2 is not None:            # Wan2.2: low-noise expert (`transformer_2`), same architecture
            if boundary_ratio is None:
                raise ValueError("mi355_flow: a two-expert Wan2.2 adapter needs `boundary_ratio` (pipeline.config.boundary_ratio)")
            self.engine_2 = WanEngine(config or WanConfig())
            self.engine_2.bind_state_dict(state_dict_2)
            self.engine_2.ready()
            self.boundary_ratio = float(boundary_ratio)
        self._video_decode = video_decode
        self.vae_decoder = None
        self.vae_max_batch = vae_max_batch
        if vae_state_dict is not None:
            from .vae import WanVAEConfig, WanVAEDecoder
            self.vae_decoder = WanVAEDecoder(vae_config or WanVAEConfig())
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
        raise RuntimeError("mi355_flow standalone adapter has no text encoder: pass prompt_embeds (and negative_prompt_embeds for CFG)")

    def decode_latents(self, latents: torch.Tensor, output_type: str = "pt"):
        """wan2_t2v.py:215-230: de-normalise, vae.decode, postprocess_video -> (B, F, 3, H, W) in [0, 1]."""
        if self.vae_decoder is not None:
            if output_type not in ("pt", "np"):
                raise ValueError("mi355_flow standalone adapter decodes to 'pt' or 'np'")
            vid = self.vae_decoder.decode(latents, postprocess=True, out_dtype=torch.float32, max_batch=self.vae_max_batch)
            return vid if output_type == "pt" else vid.permute(0, 1, 3, 4, 2).cpu().numpy()
        return self._video_decode(latents) if self._video_decode is not None else None
