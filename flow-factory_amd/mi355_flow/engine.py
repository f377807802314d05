"""Thin torch-facing wrappers over the C ABI: tensors in, tensors out, current HIP stream.

PyTorch is plumbing here (device memory, streams); every computation on the rollout hot path
runs in libmi355flow.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import BF16, DYNAMICS, F16, F32, ModelCfg

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise ValueError(f"mi355_flow: unsupported dtype {dt} (float32 / bfloat16 / float16 only)") from None


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("mi355_flow: expected a tensor on the GPU (there is no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError("mi355_flow: tensors passed to the engine must be contiguous")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


@dataclass
class TransformerConfig:
    """diffusers SD3Transformer2DModel config fields the engine needs (SD3.5-medium defaults)."""
    in_channels: int = 16
    out_channels: int = 16
    patch_size: int = 2
    num_layers: int = 24
    num_heads: int = 24
    head_dim: int = 64
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 2048
    pos_embed_max_size: int = 384
    dual_layers: Tuple[int, ...] = tuple(range(13))
    time_proj_dim: int = 256
    ff_mult: int = 4
    eps: float = 1e-6

    @property
    def dim(self) -> int:
        return self.num_heads * self.head_dim

    @classmethod
    def from_hf(cls, config) -> "TransformerConfig":
        """From a diffusers `SD3Transformer2DModel.config` (what `SD3_5Adapter` loads, reference sd3_5.py:60-66).  The engine implements
        the SD3.5 block (RMS q/k norm per head, head_dim 64, caption projection to the model width); any other member of the SD3 family
        -- e.g. SD3.0-medium, which has no q/k norm -- is refused HERE, by name, instead of failing later on a missing weight."""
        g = (lambda k, d=None: getattr(config, k, d)) if not isinstance(config, dict) else (lambda k, d=None: config.get(k, d))
        heads, hd = int(g("num_attention_heads", 24)), int(g("attention_head_dim", 64))
        qk = g("qk_norm", "rms_norm")
        if qk != "rms_norm":
            raise NotImplementedError(f"mi355_flow: SD3 transformer with qk_norm={qk!r}: the engine implements the SD3.5 block (qk_norm='rms_norm')")
        if hd != 64:
            raise NotImplementedError(f"mi355_flow: attention_head_dim={hd}: the SD3.5 engine's attention kernel is built for head_dim 64")
        cap = g("caption_projection_dim", heads * hd)
        if int(cap) != heads * hd:
            raise NotImplementedError(f"mi355_flow: caption_projection_dim={cap} differs from the model width {heads * hd}")
        dual = tuple(int(i) for i in (g("dual_attention_layers", ()) or ()))
        layers = int(g("num_layers", 24))
        if any(i < 0 or i >= layers for i in dual):
            raise ValueError(f"mi355_flow: dual_attention_layers {dual} outside [0, {layers})")
        return cls(in_channels=int(g("in_channels", 16)), out_channels=int(g("out_channels", None) or g("in_channels", 16)),
                   patch_size=int(g("patch_size", 2)), num_layers=layers, num_heads=heads, head_dim=hd,
                   joint_attention_dim=int(g("joint_attention_dim", 4096)), pooled_projection_dim=int(g("pooled_projection_dim", 2048)),
                   pos_embed_max_size=int(g("pos_embed_max_size", 384)), dual_layers=dual)

    def to_c(self) -> ModelCfg:
        mask = 0
        for i in self.dual_layers:
            mask |= 1 << int(i)
        return ModelCfg(self.in_channels, self.out_channels, self.patch_size, self.num_layers, self.num_heads,
                        self.head_dim, self.joint_attention_dim, self.pooled_projection_dim, self.pos_embed_max_size,
                        self.time_proj_dim, self.ff_mult, mask, self.eps)


class WeightHolder:
    """Shared weight-binding half of the four engines (`mi355_engine_*`, `mi355_flux_*`, `mi355_wan_*`, `mi355_vae_*`): the C side
    owns a re-packed bf16 copy of every named parameter; binding copies / converts one torch tensor into its slot."""

    _ABI = "engine"      # C symbol infix
    _WHAT = "transformer"
    KEEPALIVE_FLUSH_BYTES = 1 << 30

    def _fn(self, suffix: str):
        return getattr(self.lib, f"mi355_{self._ABI}_{suffix}")

    def param_names(self) -> List[str]:
        names = getattr(self, "_names", None)
        if names is None:
            n = self._fn("num_params")(self._h)
            names = self._names = [self._fn("param_name")(self._h, i).decode() for i in range(n)]
        return names

    def bind_tensor(self, name: str, t: torch.Tensor) -> None:
        """Enqueue the conversion of ONE parameter on the current stream; call `finish_binding()` after the last one (the source
        may be a temporary: a merged LoRA weight, a gathered FSDP shard, a host tensor's device copy)."""
        t = t.detach()
        if not t.is_cuda:
            t = t.cuda(non_blocking=True)
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        _lib.check(self._fn("bind_weight")(self._h, name.encode(), t.data_ptr(), dtype_code(t.dtype), t.dim(), shape, _stream()),
                   f"{self._ABI}_bind_weight({name})")
        self._keepalive = getattr(self, "_keepalive", [])
        self._keepalive.append(t)
        # temporaries (gathered FSDP2 shards, merged fp32 LoRA weights) must outlive their asynchronous conversion kernel, but not each
        # other: drain every ~1 GiB so that a 41 GB model never sits next to a whole un-sharded copy of itself during a re-bind
        self._keepalive_bytes = getattr(self, "_keepalive_bytes", 0) + t.numel() * t.element_size()
        if self._keepalive_bytes >= self.KEEPALIVE_FLUSH_BYTES:
            torch.cuda.current_stream().synchronize()
            self._keepalive, self._keepalive_bytes = [], 0

    def finish_binding(self) -> None:
        # conversion kernels read the (possibly temporary) source tensors asynchronously
        torch.cuda.current_stream().synchronize()
        self._keepalive, self._keepalive_bytes = [], 0

    def bind_state_dict(self, state_dict: Dict[str, torch.Tensor], partial: bool = False, strict: Optional[bool] = None) -> None:
        """Copy / re-pack torch Parameters (HF names) into the engine.  Call again after every optimizer step, EMA swap or LoRA
        merge: the weights are live during GRPO.  A state dict that lacks any expected name RAISES unless `partial=True`
        (a refresh with foreign key names -- e.g. un-merged peft keys -- must never leave stale weights bound silently;
        reference fail-fast rule, constraints.md:144-145).  `strict` is the deprecated inverse of `partial`."""
        if strict is not None:
            partial = not strict
        names = self.param_names()
        missing = [n for n in names if n not in state_dict]
        if missing and not partial:
            raise KeyError(f"mi355_flow: {self._WHAT} state dict lacks {len(missing)} of {len(names)} parameters, first: {missing[0]}"
                           + (" (peft-wrapped keys? bind through mi355_flow.binding.LiveWeights, which merges LoRA deltas)"
                              if any("base_layer" in k or "lora_" in k for k in state_dict) else ""))
        for n in names:
            if n in state_dict:
                self.bind_tensor(n, state_dict[n])
        self.finish_binding()

    def ready(self) -> None:
        _lib.check(self._fn("weights_ready")(self._h), f"{self._ABI}_weights_ready")


class Engine(WeightHolder):
    """Owns the packed bf16 copy of the transformer weights (mi355_engine)."""

    def __init__(self, cfg: TransformerConfig):
        self.lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        c = cfg.to_c()
        _lib.check(self.lib.mi355_engine_create(C.byref(c), C.byref(h)), "engine_create")
        self._h = h
        self._plans: Dict[tuple, "Plan"] = {}
        self.train_scope = False

    # ---------------------------------------------------------------- weight gradients (mi355_flow/autograd.py)
    def grad_supported(self, name: str) -> int:
        """0 = no gradient for this parameter, 1 = in the default scope (the blocks' linear layers), 2 = in the full scope only."""
        return {0: 1, 2: 2}.get(self.lib.mi355_engine_grad_supported(self._h, name.encode()), 0)

    def set_train_scope(self, full: bool) -> None:
        _lib.check(self.lib.mi355_engine_set_train_scope(self._h, int(bool(full))), "set_train_scope")
        self.train_scope = bool(full)

    def set_grad(self, name: str, grad: torch.Tensor) -> None:
        """Register the buffer the next backward writes d loss / d `name` into (same shape as the parameter): fp32, or bf16 for the weights /
        biases of the linear layers inside the blocks (`grad_supported(name) == 1`) -- the engine then rounds its fp32 sums to bf16 itself."""
        if grad.dtype not in (torch.float32, torch.bfloat16) or not grad.is_contiguous():
            raise ValueError("mi355_flow: gradient buffers are contiguous fp32 (or bf16) tensors")
        _lib.check(self.lib.mi355_engine_set_grad_typed(self._h, name.encode(), _ptr(grad), dtype_code(grad.dtype)), f"set_grad({name})")

    def clear_grads(self) -> None:
        _lib.check(self.lib.mi355_engine_clear_grads(self._h), "clear_grads")

    def attention_info(self) -> Dict[str, float]:
        """{'static': launches of a forward on the static-bound softmax kernel, 'total': attention launches, 'max_bound': ...}"""
        ns, nt, mb = C.c_int(), C.c_int(), C.c_float()
        _lib.check(self.lib.mi355_engine_attention_info(self._h, _stream(), C.byref(ns), C.byref(nt), C.byref(mb)), "attention_info")
        return {"static": ns.value, "total": nt.value, "max_bound": mb.value}

    def plan(self, batch: int, n_cfg: int, latent_h: int, latent_w: int, n_text: int, max_steps: int) -> "Plan":
        key = (batch, n_cfg, latent_h, latent_w, n_text)
        p = self._plans.get(key)
        if p is None or p.max_steps < max_steps:
            if p is not None:
                p.close()
            p = Plan(self, batch, n_cfg, latent_h, latent_w, n_text, max_steps)
            self._plans[key] = p
        return p

    def close(self) -> None:
        for p in self._plans.values():
            p.close()
        self._plans.clear()
        if self._h:
            self.lib.mi355_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Workspace for one (batch, n_cfg, latent_h, latent_w, n_text) shape (mi355_plan)."""

    def __init__(self, engine: Engine, batch: int, n_cfg: int, latent_h: int, latent_w: int, n_text: int, max_steps: int):
        self.engine, self.lib = engine, engine.lib
        self.batch, self.n_cfg, self.h, self.w, self.n_text, self.max_steps = batch, n_cfg, latent_h, latent_w, n_text, max_steps
        self.C = engine.cfg.in_channels
        h = C.c_void_p()
        _lib.check(self.lib.mi355_plan_create(engine._h, batch, n_cfg, latent_h, latent_w, n_text, max_steps, C.byref(h)),
                   "plan_create")
        self._h = h

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.mi355_plan_workspace_bytes(self._h))

    def close(self) -> None:
        if self._h:
            self.lib.mi355_plan_destroy(self._h)
            self._h = None

    # ---------------------------------------------------------------- denoiser forward
    def transformer_forward(self, latents, timestep, enc_a, pooled_a, enc_b=None, pooled_b=None, t_round_dtype=None):
        B, Bp = self.batch, self.batch * self.n_cfg
        assert latents.shape == (B, self.C, self.h, self.w), latents.shape
        t = timestep.to(device=latents.device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(Bp)
        t = t.contiguous()
        assert t.numel() == Bp
        out = torch.empty((Bp, self.C, self.h, self.w), device=latents.device, dtype=torch.bfloat16)
        rd = dtype_code(t_round_dtype if t_round_dtype is not None else latents.dtype)
        enc_a, pooled_a = _bf16c(enc_a), _bf16c(pooled_a)
        enc_b = _bf16c(enc_b) if enc_b is not None else None
        pooled_b = _bf16c(pooled_b) if pooled_b is not None else None
        latents = latents.contiguous()
        _lib.check(self.lib.mi355_transformer_forward(self._h, _stream(), _ptr(latents), dtype_code(latents.dtype), _ptr(t), rd,
                                                      _ptr(enc_a), _ptr(pooled_a), _ptr(enc_b), _ptr(pooled_b), _ptr(out)),
                   "transformer_forward")
        return out

    # ---------------------------------------------------------------- one denoise step (forward + CFG + SDE step)
    def denoise_step(self, latents, timestep, enc_a, pooled_a, enc_b, pooled_b, guidance, sigma, sigma_next, eta,
                     sigma_max, dynamics: str, noise=None, next_latents=None, compute_log_prob=True, want=()):
        B = self.batch
        dev = latents.device
        latents = latents.contiguous()
        n = latents[0].numel()
        t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B * self.n_cfg)
        elif t.numel() == B and self.n_cfg == 2:
            t = t.repeat(2)
        t = t.contiguous()
        sig, sig_n, et, stride = _scalars(sigma, sigma_next, eta, B, dev)
        outs = _StepOutputs(B, latents, want, compute_log_prob)
        enc_a, pooled_a = _bf16c(enc_a), _bf16c(pooled_a)
        enc_b = _bf16c(enc_b) if enc_b is not None else None
        pooled_b = _bf16c(pooled_b) if pooled_b is not None else None
        noise = noise.contiguous() if noise is not None else None
        nxt_in = next_latents.contiguous() if next_latents is not None else None
        _lib.check(self.lib.mi355_denoise_step(
            self._h, _stream(), _ptr(latents), dtype_code(latents.dtype), _ptr(t), _ptr(enc_a), _ptr(pooled_a),
            _ptr(enc_b), _ptr(pooled_b), float(guidance), _ptr(noise), _ptr(nxt_in),
            dtype_code(nxt_in.dtype) if nxt_in is not None else 0, _ptr(sig), _ptr(sig_n), _ptr(et), stride,
            float(sigma_max), DYNAMICS[dynamics], int(bool(compute_log_prob)), *outs.ptrs()), "denoise_step")
        return outs

    # ---------------------------------------------------------------- differentiable replay step (optimize())
    def denoise_step_train(self, latents, timestep, enc_a, pooled_a, enc_b, pooled_b, guidance, sigma, sigma_next, eta, sigma_max,
                           dynamics: str, next_latents, compute_log_prob=True, _keep=None):
        """mi355_denoise_step_train: the replay forward on per-block activation buffers (bit-identical outputs to `denoise_step`).
        `_keep` (a dict) receives the prepared device tensors so that `denoise_step_backward` can pass the same pointers."""
        B, dev = self.batch, latents.device
        latents = latents.contiguous()
        t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B * self.n_cfg)
        elif t.numel() == B and self.n_cfg == 2:
            t = t.repeat(2)
        t = t.contiguous()
        sig, sig_n, et, stride = _scalars(sigma, sigma_next, eta, B, dev)
        outs = _StepOutputs(B, latents, ("next_latents_mean", "noise_pred", "std_dev_t", "dt"), True)
        enc_a, pooled_a = _bf16c(enc_a), _bf16c(pooled_a)
        enc_b = _bf16c(enc_b) if enc_b is not None else None
        pooled_b = _bf16c(pooled_b) if pooled_b is not None else None
        nxt = next_latents.contiguous()
        k = dict(latents=latents, t=t, enc_a=enc_a, pooled_a=pooled_a, enc_b=enc_b, pooled_b=pooled_b, nxt=nxt, sig=sig, sig_n=sig_n, et=et,
                 stride=stride, guidance=float(guidance), sigma_max=float(sigma_max), dynamics=DYNAMICS[dynamics],
                 clp=int(bool(compute_log_prob)), scope=self.engine.train_scope)
        self._train_forward(k, outs)
        if _keep is not None:
            _keep.update(k)
        return outs

    def _train_forward(self, k: dict, outs) -> None:
        """One launch of mi355_denoise_step_train on prepared tensors.  The plan owns ONE activation stash (per-block buffers, modulation
        rows, LSE, ...), overwritten by every training forward: each forward takes a serial number that the matching backward checks."""
        _lib.check(self.lib.mi355_denoise_step_train(
            self._h, _stream(), _ptr(k["latents"]), dtype_code(k["latents"].dtype), _ptr(k["t"]), _ptr(k["enc_a"]), _ptr(k["pooled_a"]),
            _ptr(k["enc_b"]), _ptr(k["pooled_b"]), k["guidance"], _ptr(k["nxt"]), dtype_code(k["nxt"].dtype), _ptr(k["sig"]), _ptr(k["sig_n"]),
            _ptr(k["et"]), k["stride"], k["sigma_max"], k["dynamics"], k["clp"], None, _ptr(outs.next_latents_mean), _ptr(outs.noise_pred),
            _ptr(outs.log_prob), _ptr(outs.std_dev_t), _ptr(outs.dt)), "denoise_step_train")
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        k["serial"] = self._train_serial

    def denoise_step_backward(self, call: dict, g_log_prob, g_noise_pred, g_mean) -> None:
        """mi355_denoise_step_backward for the step recorded in `call['_keep']`; gradients go to the buffers registered with
        `Engine.set_grad`."""
        k = call["_keep"]
        if k["serial"] != getattr(self, "_train_serial", 0):
            # another training forward ran on this plan since (DPO's chosen / rejected pair, trainers/dpo.py:587-588; any loss that sums
            # several grad forwards before one backward): the stash holds ITS activations.  Re-run this step's forward on the kept
            # inputs -- same kernels, same weights (the caller re-bound them), bit-identical stash -- then differentiate it.
            self.engine.set_train_scope(k["scope"])
            self.recomputed_forwards = getattr(self, "recomputed_forwards", 0) + 1
            self._train_forward(k, _StepOutputs(self.batch, k["latents"], ("next_latents_mean", "noise_pred", "std_dev_t", "dt"), True))
        f32 = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_lp, g_np, g_mn = f32(g_log_prob), f32(g_noise_pred), f32(g_mean)
        _lib.check(self.lib.mi355_denoise_step_backward(
            self._h, _stream(), _ptr(k["latents"]), dtype_code(k["latents"].dtype), k["guidance"], _ptr(k["nxt"]), dtype_code(k["nxt"].dtype),
            _ptr(k["sig"]), _ptr(k["sig_n"]), _ptr(k["et"]), k["stride"], k["sigma_max"], k["dynamics"], k["clp"], _ptr(g_lp), _ptr(g_np),
            _ptr(g_mn)), "denoise_step_backward")

    @property
    def training_bytes(self) -> int:
        return int(self.lib.mi355_plan_training_bytes(self._h))

    # ---------------------------------------------------------------- whole rollout
    def rollout(self, timesteps: Sequence[float], sigmas: Sequence[float], noise_levels: Sequence[float], dynamics: str,
                guidance: float, init_latents: torch.Tensor, storage_dtype: torch.dtype, step_noise: torch.Tensor,
                prompt_embeds, pooled, neg_embeds=None, neg_pooled=None, keep_positions: Optional[Sequence[int]] = None,
                compute_log_prob: bool = True):
        """Returns (kept_latents [n_kept,B,C,h,w] storage dtype, log_probs [N,B] fp32 (nan where not
        computed), final_latents [B,C,h,w])."""
        N, B = len(timesteps), self.batch
        assert len(sigmas) == N + 1 and len(noise_levels) == N
        dev = init_latents.device
        keep = list(range(N + 1)) if keep_positions is None else sorted(set(int(k) for k in keep_positions))
        slots = [-1] * (N + 1)
        for s, pos in enumerate(keep):
            slots[pos] = s
        shape = (B, self.C, self.h, self.w)
        out_lat = torch.empty((len(keep),) + shape, device=dev, dtype=storage_dtype)
        out_lp = torch.full((N, B), float("nan"), device=dev, dtype=torch.float32)
        out_fin = torch.empty(shape, device=dev, dtype=storage_dtype)
        fa = C.c_float * N
        ts_c, nl_c = fa(*[float(t) for t in timesteps]), fa(*[float(e) for e in noise_levels])
        sg_c = (C.c_float * (N + 1))(*[float(s) for s in sigmas])
        sl_c = (C.c_int32 * (N + 1))(*slots)
        init_latents = init_latents.contiguous()
        step_noise = step_noise.contiguous() if step_noise is not None else None
        if step_noise is not None:
            assert step_noise.dtype == torch.float32 and step_noise.shape == (N,) + shape, step_noise.shape
        pe, pp = _bf16c(prompt_embeds), _bf16c(pooled)
        ne = _bf16c(neg_embeds) if neg_embeds is not None else None
        npl = _bf16c(neg_pooled) if neg_pooled is not None else None
        _lib.check(self.lib.mi355_rollout(
            self._h, _stream(), N, ts_c, sg_c, nl_c, DYNAMICS[dynamics], float(guidance), _ptr(init_latents),
            dtype_code(init_latents.dtype), dtype_code(storage_dtype), _ptr(step_noise), _ptr(pe), _ptr(pp), _ptr(ne),
            _ptr(npl), sl_c, _ptr(out_lat), _ptr(out_lp), _ptr(out_fin), int(bool(compute_log_prob))), "rollout")
        return out_lat, out_lp, out_fin


def _bf16c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).contiguous()


def _scalars(sigma, sigma_next, eta, B, dev):
    def one(v):
        if not isinstance(v, torch.Tensor):
            v = torch.tensor([float(v)], dtype=torch.float32)
        return v.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()

    s, sn, e = one(sigma), one(sigma_next), one(eta)
    per = max(s.numel(), sn.numel(), e.numel())
    if per == 1:
        return s, sn, e, 0
    assert per == B, "per-sample scalars must have one value per sample"
    ex = lambda v: (v.expand(B) if v.numel() == 1 else v).contiguous()
    return ex(s), ex(sn), ex(e), 1


class _StepOutputs:
    """Output buffers of one SDE step, allocated by torch (the engine never owns outputs)."""

    FIELDS = ("next_latents", "next_latents_mean", "noise_pred", "log_prob", "std_dev_t", "dt")

    def __init__(self, B: int, latents: torch.Tensor, want: Iterable[str], compute_log_prob: bool):
        dev, shp = latents.device, latents.shape
        want = set(want)
        self.next_storage = torch.empty_like(latents)
        self.next_latents = torch.empty(shp, device=dev, dtype=torch.float32) if "next_latents" in want else None
        self.next_latents_mean = torch.empty(shp, device=dev, dtype=torch.float32) if "next_latents_mean" in want else None
        self.noise_pred = torch.empty(shp, device=dev, dtype=torch.float32) if "noise_pred" in want else None
        self.log_prob = torch.empty((B,), device=dev, dtype=torch.float32) if compute_log_prob else None
        self.std_dev_t = torch.empty((B,), device=dev, dtype=torch.float32) if "std_dev_t" in want else None
        self.dt = torch.empty((B,), device=dev, dtype=torch.float32) if "dt" in want else None

    def ptrs(self):
        return (_ptr(self.next_storage), _ptr(self.next_latents), _ptr(self.next_latents_mean), _ptr(self.noise_pred),
                _ptr(self.log_prob), _ptr(self.std_dev_t), _ptr(self.dt))


def sde_step(v_text, v_uncond, guidance, latents, sigma, sigma_next, eta, sigma_max, dynamics: str, noise=None,
             next_latents=None, compute_log_prob=True, want=("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt")):
    """Standalone fused CFG + SDE/ODE step + log-prob (mi355_sde_step)."""
    lib = _lib.load()
    B = latents.shape[0]
    latents = latents.contiguous()
    n = latents[0].numel()
    sig, sig_n, et, stride = _scalars(sigma, sigma_next, eta, B, latents.device)
    outs = _StepOutputs(B, latents, want, compute_log_prob)
    # the prediction keeps its dtype (reference: `noise_pred.float()`, flow_match_euler_discrete.py:310): no silent bf16 rounding
    v_text = v_text.contiguous()
    vdt = dtype_code(v_text.dtype)
    v_uncond = v_uncond.to(v_text.dtype).contiguous() if v_uncond is not None else None
    noise = noise.to(torch.float32).contiguous() if noise is not None else None
    nxt_in = next_latents.contiguous() if next_latents is not None else None
    _lib.check(lib.mi355_sde_step(
        _stream(), B, n, _ptr(v_text), _ptr(v_uncond), vdt, float(guidance), _ptr(latents), dtype_code(latents.dtype), _ptr(noise),
        _ptr(nxt_in), dtype_code(nxt_in.dtype) if nxt_in is not None else 0, _ptr(sig), _ptr(sig_n), _ptr(et), stride,
        float(sigma_max), DYNAMICS[dynamics], int(bool(compute_log_prob)), *outs.ptrs()), "sde_step")
    return outs


def sde_step_bwd(v_text, v_uncond, guidance, latents, next_latents, sigma, sigma_next, eta, sigma_max, dynamics: str, compute_log_prob: bool,
                 g_log_prob=None, g_noise_pred=None, g_mean=None) -> torch.Tensor:
    """Adjoint of `sde_step` w.r.t. the network prediction(s) (mi355_sde_step_bwd): upstream gradients of (log_prob [B], noise_pred,
    next_latents_mean) -> d v, fp32, shape [n_cfg * B, ...] in the order [uncond, text].  `v_text` / `v_uncond`: the bf16 predictions the
    forward step consumed; `next_latents`: the stored next state of the replay."""
    lib = _lib.load()
    B = latents.shape[0]
    latents = latents.contiguous()
    n = latents[0].numel()
    if v_text.dtype != torch.bfloat16:
        raise ValueError("mi355_flow: sde_step_bwd differentiates the engine's own bf16 prediction")
    v_text = v_text.contiguous()
    v_uncond = v_uncond.contiguous() if v_uncond is not None else None
    nxt = next_latents.contiguous()
    sig, sig_n, et, stride = _scalars(sigma, sigma_next, eta, B, latents.device)
    f32 = lambda g: None if g is None else g.to(torch.float32).contiguous()
    g_lp, g_np, g_mn = f32(g_log_prob), f32(g_noise_pred), f32(g_mean)
    dv = torch.empty(((2 if v_uncond is not None else 1) * B,) + tuple(latents.shape[1:]), device=latents.device, dtype=torch.float32)
    _lib.check(lib.mi355_sde_step_bwd(
        _stream(), B, n, _ptr(v_text), _ptr(v_uncond), float(guidance), _ptr(latents), dtype_code(latents.dtype), _ptr(nxt), dtype_code(nxt.dtype),
        _ptr(sig), _ptr(sig_n), _ptr(et), stride, float(sigma_max), DYNAMICS[dynamics], int(bool(compute_log_prob)), _ptr(g_lp), _ptr(g_np),
        _ptr(g_mn), _ptr(dv)), "sde_step_bwd")
    return dv


# ---------------------------------------------------------------------- operator-level wrappers (tests / profiling)
def op_linear(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, act: int = 0) -> torch.Tensor:
    lib = _lib.load()
    x, w = _bf16c(x), _bf16c(w)
    bias = bias.to(torch.float32).contiguous()
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), device=x.device, dtype=torch.bfloat16)
    _lib.check(lib.mi355_op_linear(_stream(), _ptr(x), _ptr(w), _ptr(bias), _ptr(out), M, N, K, act), "op_linear")
    return out


def op_linear_gate_res(x: torch.Tensor, a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, gate: torch.Tensor, rows_per_sample: int) -> torch.Tensor:
    """x (bf16 [M, N], updated IN PLACE and returned) += gate[m // rows_per_sample] * (a @ w.T + bias)."""
    lib = _lib.load()
    a, w, gate = _bf16c(a), _bf16c(w), _bf16c(gate)
    bias = bias.to(torch.float32).contiguous()
    M, K = a.shape
    N = w.shape[0]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape == (M, N) and gate.shape == ((M + rows_per_sample - 1) // rows_per_sample, N)
    _lib.check(lib.mi355_op_linear_gate_res(_stream(), _ptr(a), _ptr(w), _ptr(bias), _ptr(gate), _ptr(x), M, N, K, rows_per_sample),
               "op_linear_gate_res")
    return x


def op_wgrad(dy: torch.Tensor, x: torch.Tensor, k_split: int = 1, variant: int = 1, want_colsum: bool = False):
    """Weight-gradient partial sums [k_split, N, K] (fp32) of dW = dy^T @ x over `k_split` slices of the M rows; dy [M, N] and x [M, K] bf16
    row-major (rows may be strided views: the row stride is passed; any M: a ragged last 64-row tile reads zeros).  variant 1 = the
    row-major-operand kernel (csrc/gemm_tn.hip), 2 = its 256 x 256-tile form, 0 = the transposed-copy path; all return the same bits.  `want_colsum` (variant 1): also the per-slice column sums of dy [k_split, N] the kernel
    takes from the fragments it holds (the bias gradient's partials)."""
    lib = _lib.load()
    assert dy.is_cuda and x.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.stride(1) == 1 and x.stride(1) == 1
    assert dy.shape[0] == x.shape[0]
    M, N = dy.shape
    K = x.shape[1]
    out = torch.empty((k_split, N, K), device=dy.device, dtype=torch.float32)
    scratch = torch.empty(((N + K) * ((M + 63) // 64 * 64),), device=dy.device, dtype=torch.bfloat16) if variant == 0 else None
    C = __import__("ctypes")
    colsum = torch.zeros((k_split, N), device=dy.device, dtype=torch.float32) if want_colsum else None
    _lib.check(lib.mi355_op_wgrad(_stream(), C.c_void_p(dy.data_ptr()), dy.stride(0), C.c_void_p(x.data_ptr()), x.stride(0), _ptr(out), M, N, K, k_split, variant,
                                  _ptr(scratch) if scratch is not None else None, _ptr(colsum) if colsum is not None else None), "op_wgrad")
    return (out, colsum) if want_colsum else out


def op_attention(q: torch.Tensor, k: torch.Tensor, vT: torch.Tensor, S: int, n_img: int):
    """q,k: [B,H,S_pad,64] bf16; vT: [B,H,64,S_pad] bf16 -> (o_img [B*n_img, H*64], o_ctx [B*(S-n_img), H*64])."""
    lib = _lib.load()
    B, H, S_pad, hd = q.shape
    assert hd == 64 and vT.shape == (B, H, 64, S_pad)
    o_img = torch.empty((B * n_img, H * 64), device=q.device, dtype=torch.bfloat16)
    o_ctx = torch.empty((max(B * (S - n_img), 1), H * 64), device=q.device, dtype=torch.bfloat16)
    _lib.check(lib.mi355_op_attention(_stream(), _ptr(q.contiguous()), _ptr(k.contiguous()), _ptr(vT.contiguous()),
                                      _ptr(o_img), _ptr(o_ctx), B, H, S, S_pad, n_img), "op_attention")
    return o_img, o_ctx[: B * (S - n_img)]


def op_ln_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, rows_per_sample: int, eps: float = 1e-6):
    lib = _lib.load()
    M, D = x.shape
    mod = torch.stack([_bf16c(shift), _bf16c(scale)], 0).contiguous()  # [2][nb][D]: one allocation
    out = torch.empty_like(x)
    _lib.check(lib.mi355_op_ln_modulate(_stream(), _ptr(_bf16c(x)), _ptr(mod[0]), _ptr(mod[1]), _ptr(out), M, D,
                                        rows_per_sample, eps), "op_ln_modulate")
    return out
