"""Qwen-Image rollout on the native engine (mi355_qwen_*): host mirror of `QwenImageAdapter.inference` / `.forward`
(reference src/flow_factory/models/qwen_image/qwen_image.py:288-438, :476-600) -- SURVEY.md 8(f) row N4 (config E).

Same contract as the SD3.5 / FLUX paths: reference argument names and defaults, the reference's RNG draw order, trajectory /
log-prob / callback collectors, no CPU or PyTorch fallback.  Qwen-Image specifics kept from the reference: packed latents
`(B, h/2*w/2, 64)`, `timestep = t.to(latents.dtype) / 1000`, ragged prompts (`prompt_embeds_mask`, `_pad_batch_prompt`), true CFG
with a negative prompt of its own length and the norm rescale of qwen_image.py:579-587.

What is different on MI355X: the 41 GB of bf16 weights stay resident (no FSDP2 gather per step: the reference shards the 20 B
parameters over 80 GB GPUs, one 288 GB MI355X holds model + workspace); the cond and uncond evaluations run as ONE forward batch
`[negative | positive]`; padded text keys are masked inside the attention kernel, so one launch serves prompts of any length.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from ._lib import DYNAMICS, QwenCfg
from .engine import WeightHolder, _bf16c, _ptr, _stream, dtype_code, sde_step
from .flux import _DTYPE_MAP, VAE_SCALE_FACTOR, pack_latents, unpack_latents
from .samples import QwenImageSample
from .scheduler import (FlowMatchEulerDiscreteSDEScheduler, SDESchedulerOutput, host_noise_levels, randn_tensor,
                        set_scheduler_timesteps)
from .trajectory import TrajectoryIndicesType, _resolve, collect_rollout

TEXT_PAD = 32     # plans are keyed by the padded text length: round up so that near-equal prompt lengths share one workspace


@dataclass
class QwenConfig:
    """diffusers QwenImageTransformer2DModel config fields the engine needs (Qwen-Image defaults)."""
    in_channels: int = 64
    num_layers: int = 60
    num_attention_heads: int = 24
    attention_head_dim: int = 128
    joint_attention_dim: int = 3584
    axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)
    scale_rope: bool = True
    time_proj_dim: int = 256
    eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    def to_c(self) -> QwenCfg:
        return QwenCfg(self.in_channels, self.num_layers, self.num_attention_heads, self.attention_head_dim, self.joint_attention_dim,
                       self.time_proj_dim, (C.c_int32 * 3)(*self.axes_dims_rope), int(self.scale_rope), self.eps)


def model_timestep(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """The angle base of the sinusoidal projection: `timestep = t.to(latents.dtype)` (qwen_image.py:497), `timestep / 1000` in that
    dtype (:534), then `Timesteps(scale=1000)` multiplies by 1000 in fp32."""
    return (t.float().to(dtype) / 1000).float() * 1000.0


def pad_batch_prompt(prompt_embeds_mask, prompt_embeds, device):
    """`QwenImageAdapter._pad_batch_prompt` (qwen_image.py:238-284): lists of ragged tensors or padded batches -> (txt_seq_lens,
    mask (B, L), embeds (B, L, J)) truncated to the longest valid prompt L of the batch."""
    if isinstance(prompt_embeds_mask, (list, tuple)):
        lens = [int(m.sum()) for m in prompt_embeds_mask]
        mask = torch.nn.utils.rnn.pad_sequence([m.to(device) for m in prompt_embeds_mask], batch_first=True, padding_value=0)
    else:
        mask = prompt_embeds_mask.to(device)
        lens = [int(v) for v in mask.sum(dim=1).tolist()]
    L = max(lens)
    if isinstance(prompt_embeds, (list, tuple)):
        emb = torch.nn.utils.rnn.pad_sequence([e.to(device) for e in prompt_embeds], batch_first=True, padding_value=0.0)
    else:
        emb = prompt_embeds.to(device)
    return lens, mask[:, :L], emb[:, :L]


class QwenEngine(WeightHolder):
    """Owns the packed bf16 copy of the Qwen-Image transformer weights (mi355_qwen): 41 GB resident for the 60-layer model."""

    _ABI, _WHAT = "qwen", "Qwen-Image transformer"

    def __init__(self, cfg: QwenConfig = QwenConfig()):
        self.lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        c = cfg.to_c()
        _lib.check(self.lib.mi355_qwen_create(C.byref(c), C.byref(h)), "qwen_create")
        self._h = h
        self._plans: Dict[tuple, "QwenPlan"] = {}

    def plan(self, batch: int, n_cfg: int, latent_h: int, latent_w: int, n_text: int, max_steps: int) -> "QwenPlan":
        key = (batch, n_cfg, latent_h, latent_w, n_text)
        p = self._plans.get(key)
        if p is None or p.max_steps < max_steps:
            if p is not None:
                p.close()
            p = QwenPlan(self, batch, n_cfg, latent_h, latent_w, n_text, max_steps)
            self._plans[key] = p
        return p

    # ---------------------------------------------------------------- weight gradients (mi355_flow/autograd.py: qwen_replay)
    def grad_supported(self, name: str) -> int:
        """1 = the native backward produces a gradient for this parameter (the linear layers inside the transformer blocks), 0 = it does not."""
        return 1 if self.lib.mi355_qwen_grad_supported(self._h, name.encode()) == 0 else 0

    def set_grad(self, name: str, grad: torch.Tensor) -> None:
        """Register the buffer the next backward writes d loss / d `name` into (same shape as the parameter): fp32, or bf16 for the weights /
        biases of the linear layers inside the blocks (`grad_supported(name) == 1`) -- the engine then rounds its fp32 sums to bf16 itself."""
        if grad.dtype not in (torch.float32, torch.bfloat16) or not grad.is_contiguous():
            raise ValueError("mi355_flow: gradient buffers are contiguous fp32 (or bf16) tensors")
        _lib.check(self.lib.mi355_qwen_set_grad_typed(self._h, name.encode(), _ptr(grad), dtype_code(grad.dtype)), f"qwen_set_grad({name})")

    def clear_grads(self) -> None:
        _lib.check(self.lib.mi355_qwen_clear_grads(self._h), "qwen_clear_grads")

    def close(self) -> None:
        for p in self._plans.values():
            p.close()
        self._plans.clear()
        if self._h:
            self.lib.mi355_qwen_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QwenPlan:
    def __init__(self, engine: QwenEngine, batch: int, n_cfg: int, latent_h: int, latent_w: int, n_text: int, max_steps: int):
        self.engine, self.lib = engine, engine.lib
        self.batch, self.n_cfg, self.h, self.w, self.n_text, self.max_steps = batch, n_cfg, latent_h, latent_w, n_text, max_steps
        self.Ni = (latent_h // 2) * (latent_w // 2)
        self.C = engine.cfg.in_channels
        h = C.c_void_p()
        _lib.check(self.lib.mi355_qwen_plan_create(engine._h, batch, n_cfg, latent_h, latent_w, n_text, max_steps, C.byref(h)), "qwen_plan_create")
        self._h = h

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.mi355_qwen_plan_workspace_bytes(self._h))

    def close(self) -> None:
        if self._h:
            self.lib.mi355_qwen_plan_destroy(self._h)
            self._h = None

    def _text(self, embeds: torch.Tensor, lens: Optional[Sequence[int]]):
        FB = self.batch * self.n_cfg
        assert embeds.shape == (FB, self.n_text, self.engine.cfg.joint_attention_dim), (embeds.shape, FB, self.n_text)
        lens_c = None
        if lens is not None:
            assert len(lens) == FB
            lens_c = (C.c_int32 * FB)(*[int(v) for v in lens])
        return _bf16c(embeds), lens_c

    def transformer_forward(self, latents: torch.Tensor, t_model: torch.Tensor, embeds: torch.Tensor, lens: Optional[Sequence[int]] = None,
                            guidance_scale: float = 1.0, return_raw: bool = False):
        """latents (B, Ni, 64) packed; t_model (B,) = `model_timestep(t, latents.dtype)`; embeds (n_cfg*B, n_text, J), negative prompts
        first when n_cfg == 2; lens = valid text tokens per forward sample.  Returns the prediction the scheduler sees (B, Ni, 64) bf16
        (and the raw network outputs (n_cfg*B, Ni, 64) with return_raw)."""
        B = self.batch
        assert latents.shape == (B, self.Ni, self.C), latents.shape
        dev = latents.device
        tm = t_model.to(device=dev, dtype=torch.float32).reshape(-1)
        tm = (tm.expand(B) if tm.numel() == 1 else tm).contiguous()
        out = torch.empty((B, self.Ni, self.C), device=dev, dtype=torch.bfloat16)
        raw = torch.empty((B * self.n_cfg, self.Ni, self.C), device=dev, dtype=torch.bfloat16) if return_raw else None
        latents = latents.contiguous()
        pe, lens_c = self._text(embeds, lens)
        _lib.check(self.lib.mi355_qwen_forward(self._h, _stream(), _ptr(latents), dtype_code(latents.dtype), _ptr(tm), _ptr(pe), lens_c,
                                               float(guidance_scale), _ptr(out), _ptr(raw)), "qwen_forward")
        return (out, raw) if return_raw else out

    # ---------------------------------------------------------------- differentiable forward (optimize() replay)
    def forward_train(self, latents: torch.Tensor, t_model: torch.Tensor, embeds: torch.Tensor, lens: Optional[Sequence[int]] = None,
                      guidance_scale: float = 1.0) -> torch.Tensor:
        """mi355_qwen_forward_train: `transformer_forward` on per-block activation buffers -- the same kernel binaries, so the prediction is
        bit-identical -- keeping what `backward` needs in the plan's training stash (ONE per plan; every call takes a serial number)."""
        B = self.batch
        assert latents.shape == (B, self.Ni, self.C), latents.shape
        dev = latents.device
        tm = t_model.to(device=dev, dtype=torch.float32).reshape(-1)
        tm = (tm.expand(B) if tm.numel() == 1 else tm).contiguous()
        out = torch.empty((B, self.Ni, self.C), device=dev, dtype=torch.bfloat16)
        latents = latents.contiguous()
        pe, lens_c = self._text(embeds, lens)
        _lib.check(self.lib.mi355_qwen_forward_train(self._h, _stream(), _ptr(latents), dtype_code(latents.dtype), _ptr(tm), _ptr(pe), lens_c,
                                                     float(guidance_scale), _ptr(out), None), "qwen_forward_train")
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        return out

    def backward(self, dv: torch.Tensor) -> None:
        """mi355_qwen_backward: d loss / d v [B, Ni, C] fp32 of the LAST `forward_train` -> the buffers registered with `QwenEngine.set_grad`."""
        dv = dv.to(torch.float32).contiguous()
        assert dv.shape == (self.batch, self.Ni, self.C), dv.shape
        _lib.check(self.lib.mi355_qwen_backward(self._h, _stream(), _ptr(dv)), "qwen_backward")

    @property
    def training_bytes(self) -> int:
        return int(self.lib.mi355_qwen_plan_training_bytes(self._h))

    def rollout(self, timesteps: Sequence[float], sigmas: Sequence[float], noise_levels: Sequence[float], dynamics: str,
                guidance_scale: float, init_latents: torch.Tensor, storage_dtype: torch.dtype, step_noise: Optional[torch.Tensor],
                embeds: torch.Tensor, lens: Optional[Sequence[int]] = None, keep_positions: Optional[Sequence[int]] = None,
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
        pe, lens_c = self._text(embeds, lens)
        _lib.check(self.lib.mi355_qwen_rollout(
            self._h, _stream(), N, ts_c, sg_c, nl_c, DYNAMICS[dynamics], float(guidance_scale), _ptr(init_latents),
            dtype_code(init_latents.dtype), dtype_code(storage_dtype), _ptr(step_noise), _ptr(pe), lens_c, sl_c, _ptr(out_lat),
            _ptr(out_lp), _ptr(out_fin), int(bool(compute_log_prob))), "qwen_rollout")
        return out_lat, out_lp, out_fin


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class QwenRolloutMixin:
    """`inference()` / `forward()` of `QwenImageAdapter` (reference models/qwen_image/qwen_image.py:288-438, :476-600) on the engine.
    Host classes provide `engine` (QwenEngine), `scheduler`, `device`, `latent_storage_dtype`, `encode_prompt`,
    `decode_latents(latents, height, width, output_type)`.  Used by the standalone `QwenImageNativeAdapter` below and, mixed in FRONT of
    the reference's own `QwenImageAdapter`, by `mi355_flow.flow_factory_plugin.QwenImageNativeAdapter`."""

    _sample_cls = QwenImageSample
    _output_cls = SDESchedulerOutput
    _set_timesteps = staticmethod(set_scheduler_timesteps)

    def _before_engine_call(self) -> None:
        """Hook run at the top of inference() / forward(): the Flow-Factory plugin re-binds changed weights here."""

    def _check_attention_kwargs(self, kw) -> None:
        kw = dict(kw or {})
        scale = float(kw.pop("scale", 1.0))
        if kw:
            raise NotImplementedError(f"mi355_flow: attention_kwargs {sorted(kw)} are not supported by the native engine "
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

    # ------------------------------------------------------------------ text: [negative | positive], zero-padded to the plan length
    def _forward_text(self, prompt_embeds, prompt_embeds_mask, negative_prompt_embeds, negative_prompt_embeds_mask, guidance_scale, device):
        """qwen_image.py:499-528: CFG runs when guidance_scale > 1 and a negative prompt (embeds + mask) is given.  Returns
        (n_cfg, embeds (n_cfg*B, L, J), lens, (pos_lens, pos_mask, pos_embeds), (neg...) or None)."""
        has_neg = negative_prompt_embeds is not None and negative_prompt_embeds_mask is not None
        do_cfg = float(guidance_scale) > 1.0 and has_neg
        pos = pad_batch_prompt(prompt_embeds_mask, prompt_embeds, device)
        neg = pad_batch_prompt(negative_prompt_embeds_mask, negative_prompt_embeds, device) if do_cfg else None
        L = max(pos[2].shape[1], neg[2].shape[1] if neg else 0)
        L = _round_up(L, TEXT_PAD)

        def padded(e):
            out = torch.zeros((e.shape[0], L, e.shape[2]), device=device, dtype=torch.bfloat16)
            out[:, :e.shape[1]] = e
            return out

        if do_cfg:
            if neg[2].shape[0] != pos[2].shape[0]:
                raise ValueError("mi355_flow: negative_prompt_embeds must have the batch size of prompt_embeds")
            return 2, torch.cat([padded(neg[2]), padded(pos[2])], dim=0), list(neg[0]) + list(pos[0]), pos, neg
        return 1, padded(pos[2]), list(pos[0]), pos, None

    # ------------------------------------------------------------------ rollout (qwen_image.py:288-438)
    @torch.no_grad()
    def inference(
        self,
        prompt: Union[str, List[str]] = None,
        negative_prompt: Union[str, List[str]] = None,
        num_inference_steps: int = 50,
        guidance_scale: float = 4.0,
        height: int = 1024,
        width: int = 1024,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        prompt_ids: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
        prompt_embeds: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
        prompt_embeds_mask: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
        negative_prompt_ids: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
        negative_prompt_embeds: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
        negative_prompt_embeds_mask: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
        attention_kwargs: Optional[Dict[str, Any]] = {},
        max_sequence_length: int = 1024,
        compute_log_prob: bool = False,
        extra_call_back_kwargs: List[str] = [],
        trajectory_indices: TrajectoryIndicesType = "all",
    ) -> List[QwenImageSample]:
        self._before_engine_call()
        device = self.device
        self._check_attention_kwargs(attention_kwargs)
        if (prompt is not None and (prompt_embeds is None or prompt_embeds_mask is None)) or (
                negative_prompt is not None and (negative_prompt_embeds is None or negative_prompt_embeds_mask is None)):
            enc = self.encode_prompt(prompt=prompt, negative_prompt=negative_prompt, guidance_scale=guidance_scale,
                                     max_sequence_length=max_sequence_length, device=device)
            prompt_ids, prompt_embeds, prompt_embeds_mask = enc["prompt_ids"], enc["prompt_embeds"], enc["prompt_embeds_mask"]
            negative_prompt_ids = enc.get("negative_prompt_ids")
            negative_prompt_embeds = enc.get("negative_prompt_embeds")
            negative_prompt_embeds_mask = enc.get("negative_prompt_embeds_mask")
        if prompt_embeds is None or prompt_embeds_mask is None:
            raise ValueError("mi355_flow: pass `prompt` or `prompt_embeds` + `prompt_embeds_mask`")
        B = len(prompt_embeds)
        dtype = getattr(self, "transformer_dtype", torch.bfloat16)
        Cl = self.engine.cfg.in_channels // 4
        # QwenImagePipeline.prepare_latents: height = 2 * (height // (vae_scale_factor * 2)); draws (B, 1, 16, h, w) and packs it
        h = 2 * (int(height) // (VAE_SCALE_FACTOR * 2))
        w = 2 * (int(width) // (VAE_SCALE_FACTOR * 2))
        N = int(num_inference_steps)
        Ni = (h // 2) * (w // 2)
        img_shapes = [[(1, h // 2, w // 2)]] * B

        # RNG in the reference's order: the initial latents, then one fp32 tensor of the PACKED shape per step (none under ODE)
        dyn = self.scheduler.dynamics_type
        latents = pack_latents(randn_tensor((B, 1, Cl, h, w), generator=generator, device=device, dtype=dtype).reshape(B, Cl, h, w))
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
        n_cfg, embeds, lens, _, _ = self._forward_text(prompt_embeds, prompt_embeds_mask, negative_prompt_embeds, negative_prompt_embeds_mask,
                                                       guidance_scale, device)
        plan = self.engine.plan(B, n_cfg, h, w, embeds.shape[1], N)
        stepwise = any(k != "noise_level" for k in extra_call_back_kwargs)
        kept = _resolve(trajectory_indices, N + 1)
        keep_positions = list(range(N + 1)) if kept is None else sorted(kept)
        step_outputs = None
        if not stepwise:
            lat_kept, log_probs, final = plan.rollout(ts_host, sig_host, eta_host, dyn, guidance_scale, latents, storage, step_noise, embeds,
                                                      lens, keep_positions=keep_positions, compute_log_prob=compute_log_prob)
            pos_to_slot = {p: s for s, p in enumerate(keep_positions)}
        else:
            lat_kept, log_probs, step_outputs = self._rollout_stepwise(plan, ts_host, sig_host, eta_host, guidance_scale, latents, storage,
                                                                       step_noise, embeds, lens, compute_log_prob, extra_call_back_kwargs)
            final = lat_kept[N]
            pos_to_slot = {p: p for p in range(N + 1)}

        traj = collect_rollout(trajectory_indices, N, lambda pos: lat_kept[pos_to_slot[pos]], log_probs, eta_host, compute_log_prob,
                               step_outputs, extra_call_back_kwargs,
                               captured_noise_levels=host_noise_levels(self.scheduler, N, effective=False), dynamics=dyn)
        images = self.decode_latents(final, height, width, output_type="pt")
        pick = lambda v, b: v[b] if v is not None else None
        return [
            self._sample_cls(
                timesteps=timesteps,
                **traj.per_sample(b),
                height=height, width=width,
                image=images[b] if images is not None else None,
                img_shapes=img_shapes[b],
                prompt=prompt[b] if isinstance(prompt, list) else prompt,
                prompt_ids=pick(prompt_ids, b),
                prompt_embeds=prompt_embeds[b],
                prompt_embeds_mask=prompt_embeds_mask[b],
                negative_prompt=negative_prompt[b] if isinstance(negative_prompt, list) else negative_prompt,
                negative_prompt_ids=pick(negative_prompt_ids, b),
                negative_prompt_embeds=pick(negative_prompt_embeds, b),
                negative_prompt_embeds_mask=pick(negative_prompt_embeds_mask, b),
            )
            for b in range(B)
        ]

    def _rollout_stepwise(self, plan, ts, sig, eta, guidance, latents, storage, step_noise, embeds, lens, compute_log_prob, extra_keys):
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
            v = plan.transformer_forward(cur, model_timestep(f32(ts[i]).reshape(1), storage), embeds, lens, guidance)
            o = sde_step(v, None, 1.0, cur, f32(ts[i]) / f32(1000.0), f32(t_next) / f32(1000.0), eta[i], sig[1], self.scheduler.dynamics_type,
                         noise=step_noise[i] if step_noise is not None else None, compute_log_prob=clp, want=want)
            if clp:
                log_probs[i] = o.log_prob
            cur = o.next_storage
            all_lat.append(cur)
            outs.append(o)
        return all_lat, log_probs, outs

    # ------------------------------------------------------------------ single step / replay (qwen_image.py:476-600), no-grad
    def forward(
        self,
        t: torch.Tensor,
        latents: torch.Tensor,
        prompt_embeds: torch.Tensor,
        prompt_embeds_mask: torch.Tensor,
        img_shapes: List[List[Tuple[int, int, int]]],
        negative_prompt_embeds: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
        negative_prompt_embeds_mask: Optional[Union[List[torch.Tensor], torch.Tensor]] = None,
        guidance_scale: float = 4.0,
        t_next: Optional[torch.Tensor] = None,
        next_latents: Optional[torch.Tensor] = None,
        noise_level: Optional[float] = None,
        attention_kwargs: Optional[Dict[str, Any]] = None,
        compute_log_prob: bool = True,
        return_kwargs: List[str] = ["noise_pred", "next_latents", "next_latents_mean", "std_dev_t", "dt", "log_prob"],
    ) -> SDESchedulerOutput:
        kw = dict(t=t, latents=latents, prompt_embeds=prompt_embeds, prompt_embeds_mask=prompt_embeds_mask, img_shapes=img_shapes,
                  negative_prompt_embeds=negative_prompt_embeds, negative_prompt_embeds_mask=negative_prompt_embeds_mask,
                  guidance_scale=guidance_scale, t_next=t_next, next_latents=next_latents, noise_level=noise_level,
                  attention_kwargs=attention_kwargs, compute_log_prob=compute_log_prob, return_kwargs=return_kwargs)
        if torch.is_grad_enabled() and getattr(self, "_live_weights", None) is not None:
            # optimize() (trainers/grpo.py:263; trainers/dgpo.py:352-364): the replay WITH autograd on the engine's differentiable forward +
            # native backward (mi355_flow.autograd.qwen_replay) when its backward covers the trainable set
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

    def _grad_fallback(self, why: str, kwargs: Dict[str, Any]):
        """Grad-mode forward() the native backward cannot serve.  Standalone: there is no other implementation -- raise (the Flow-Factory
        plugin overrides this with the reference's autograd path)."""
        raise NotImplementedError(f"mi355_flow: Qwen-Image forward() with autograd is not available natively: {why}")

    def _forward_impl(self, t, latents, prompt_embeds, prompt_embeds_mask, img_shapes, negative_prompt_embeds, negative_prompt_embeds_mask,
                      guidance_scale, t_next, next_latents, noise_level, attention_kwargs, compute_log_prob, return_kwargs,
                      grad: bool) -> SDESchedulerOutput:
        self._before_engine_call()
        self._check_attention_kwargs(attention_kwargs)
        B, Ni, _ = latents.shape
        dev = latents.device
        shape = img_shapes[0][0] if isinstance(img_shapes[0], (list, tuple)) and isinstance(img_shapes[0][0], (list, tuple)) else img_shapes[0]
        _, hp, wp = (int(v) for v in shape)
        if hp * wp != Ni:
            raise ValueError(f"mi355_flow: img_shapes {shape} does not match {Ni} packed tokens")
        n_cfg, embeds, lens, _, _ = self._forward_text(prompt_embeds, prompt_embeds_mask, negative_prompt_embeds, negative_prompt_embeds_mask,
                                                       guidance_scale, dev)
        plan = self.engine.plan(B, n_cfg, 2 * hp, 2 * wp, embeds.shape[1], 1)
        t = torch.as_tensor(t, device=dev, dtype=torch.float32).reshape(-1)
        tm = model_timestep(t, latents.dtype)
        if not grad:
            v = plan.transformer_forward(latents, tm, embeds, lens, guidance_scale)
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
            call = dict(latents=latents, train_args=(latents, tm, embeds, lens, float(guidance_scale)), sigma=sigma, sigma_next=sigma_next,
                        eta=noise_level, sigma_max=float(sched.sigmas[1]), dynamics=dyn, next_latents=next_latents if replay else latents,
                        compute_log_prob=clp)
            lp, npred, mean, std, dtt = AG.qwen_replay(self, plan, call)
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


class QwenImageNativeAdapter(QwenRolloutMixin):
    """Standalone Qwen-Image adapter (no Flow-Factory import): engine + scheduler.  `source` is a state dict or an `nn.Module`
    (possibly FSDP2-wrapped / LoRA-wrapped: bound through `mi355_flow.binding.LiveWeights`, re-bound when its parameters change)."""

    def __init__(self, source, config: Optional[QwenConfig] = None, scheduler: Optional[FlowMatchEulerDiscreteSDEScheduler] = None,
                 latent_storage_dtype: Optional[str] = "bf16", transformer_dtype: torch.dtype = torch.bfloat16,
                 device: Union[str, torch.device] = "cuda", vae_state_dict: Optional[Dict[str, torch.Tensor]] = None, vae_config=None,
                 vae_max_batch: int = 4):
        if not torch.cuda.is_available():
            raise RuntimeError("mi355_flow: no GPU visible; the native rollout engine has no CPU path")
        self.device = torch.device(device)
        self.transformer_dtype = transformer_dtype
        self._latent_storage = latent_storage_dtype
        # Qwen-Image scheduler config: dynamic exponential shifting (mu from the image sequence length), terminal sigma 0.02
        self.scheduler = scheduler or FlowMatchEulerDiscreteSDEScheduler(
            shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, base_image_seq_len=256, max_image_seq_len=8192,
            shift_terminal=0.02, sde_steps=[1, 2, 3], num_sde_steps=1)
        self.engine = QwenEngine(config or QwenConfig())
        self.vae_decoder = None
        self.vae_max_batch = vae_max_batch
        if vae_state_dict is not None:                    # AutoencoderKLQwenImage = the causal 3-D video VAE on one frame
            from .vae import WanVAEConfig, WanVAEDecoder
            self.vae_decoder = WanVAEDecoder(vae_config or WanVAEConfig())
            self.vae_decoder.bind_state_dict(vae_state_dict)
            self.vae_decoder.ready()
        self._live_weights = None
        if isinstance(source, torch.nn.Module):
            from .binding import LiveWeights
            module = source
            self._live_weights = LiveWeights(self.engine, lambda: module)
            self._sync_weights()
        else:
            self.refresh_weights(source)

    @property
    def latent_storage_dtype(self) -> Optional[torch.dtype]:
        return _DTYPE_MAP.get(self._latent_storage) if self._latent_storage else None

    def _sync_weights(self) -> int:
        if self._live_weights is None:
            return 0
        n = self._live_weights.sync()
        if n:
            self.engine.ready()
        return n

    def _before_engine_call(self) -> None:
        self._sync_weights()

    def refresh_weights(self, state_dict: Dict[str, torch.Tensor]) -> None:
        self.engine.bind_state_dict(state_dict)
        self.engine.ready()

    def rollout(self):
        self.scheduler.rollout()

    def eval(self):
        self.scheduler.eval()

    def train(self, mode: bool = True):
        self.scheduler.train(mode)

    def encode_prompt(self, *a, **k):
        raise RuntimeError("mi355_flow standalone adapter has no text encoder: pass prompt_embeds + prompt_embeds_mask")

    def decode_latents(self, latents: torch.Tensor, height: int, width: int, output_type: str = "pt"):
        """qwen_image.py:197-213: unpack to (B, 16, 1, h, w), latents / (1/std) + mean, vae.decode(...)[:, :, 0], postprocess."""
        if self.vae_decoder is None:
            return None
        if output_type not in ("pt", "np"):
            raise ValueError("mi355_flow standalone adapter decodes to 'pt' or 'np'")
        img = decode_packed_latents(self.vae_decoder, latents, height, width, max_batch=self.vae_max_batch)
        return img if output_type == "pt" else img.float().permute(0, 2, 3, 1).cpu().numpy()


def decode_packed_latents(decoder, latents: torch.Tensor, height: int, width: int, postprocess: bool = True, max_batch: int = 4,
                          out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """Packed (B, Ni, 64) latents -> (B, 3, H, W): `_unpack_latents` to one latent frame, the video VAE decoder on it, frame 0."""
    h = 2 * (int(height) // (VAE_SCALE_FACTOR * 2))
    w = 2 * (int(width) // (VAE_SCALE_FACTOR * 2))
    lat = unpack_latents(latents, h, w).unsqueeze(2).contiguous()
    return decoder.decode(lat, postprocess=postprocess, out_dtype=out_dtype, max_batch=max_batch)[:, 0]


def op_cfg_rescale(v_neg: torch.Tensor, v_pos: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    """(rows, 64) bf16 x 2 -> the norm-rescaled true-CFG prediction (qwen_image.py:579-587)."""
    lib = _lib.load()
    a, b = _bf16c(v_neg), _bf16c(v_pos)
    out = torch.empty_like(a)
    rows = a.numel() // a.shape[-1]
    _lib.check(lib.mi355_op_cfg_rescale(_stream(), _ptr(a), _ptr(b), float(guidance_scale), _ptr(out), rows, a.shape[-1]), "op_cfg_rescale")
    return out


def op_rms_rows(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    lib = _lib.load()
    a = _bf16c(x)
    w = weight.float().contiguous()
    out = torch.empty_like(a)
    _lib.check(lib.mi355_op_rms_rows(_stream(), _ptr(a), _ptr(w), _ptr(out), a.numel() // a.shape[-1], a.shape[-1], eps), "op_rms_rows")
    return out


def op_cfg_rescale_bwd(v_neg: torch.Tensor, v_pos: torch.Tensor, guidance_scale: float, d_out: torch.Tensor):
    """Adjoint of `op_cfg_rescale` (norms differentiated through): d_out fp32 (rows, 64) -> (d_neg, d_pos) bf16."""
    lib = _lib.load()
    a, b = _bf16c(v_neg), _bf16c(v_pos)
    d = d_out.to(torch.float32).contiguous()
    dn, dp = torch.empty_like(a), torch.empty_like(a)
    rows = a.numel() // a.shape[-1]
    _lib.check(lib.mi355_op_cfg_rescale_bwd(_stream(), _ptr(a), _ptr(b), float(guidance_scale), _ptr(d), _ptr(dn), _ptr(dp), rows, a.shape[-1]),
               "op_cfg_rescale_bwd")
    return dn, dp
