"""Differentiable engine forward for the `optimize()` replay (reference
src/flow_factory/trainers/grpo.py:185-342).

`GRPOTrainer.optimize` calls `adapter.forward(t, latents, next_latents=x_{i+1}, ...)` WITH autograd, builds the PPO-clip loss
from `output.log_prob` (+ an optional KL term from `noise_pred` / `next_latents_mean`) and calls `accelerator.backward(loss)`.
`denoise_replay` below is that forward on the engine:

  * forward  = `mi355_denoise_step_train`: the launch sequence of the rollout's denoise step (same kernels, same epilogues), so
    the replay log-prob is bit-identical to the rollout's and `ratio == exp(0) == 1` before any update -- the reference's
    train/inference-consistency invariant (.agents/knowledge/topics/train_inference_consistency.md:20-29), which cannot hold to
    the default `clip_range` of +-1e-4 when the replay runs on a different implementation than the rollout;
  * backward = `mi355_denoise_step_backward`: hand-written HIP (flash-attention backward, dgrad GEMMs with fused GELU', LayerNorm /
    RMSNorm backward, split-K wgrad), fp32 weight gradients handed to torch autograd as the gradients of the (possibly
    LoRA-merged) weight tensors -- so LoRA A/B gradients, DDP's bucketed all-reduce (RCCL) and DeepSpeed's hooks all see ordinary
    `.grad` flow.  Merged LoRA weights `W + s * B @ A` are built with autograd here and bound to the engine as they are.

Supported trainable set: every parameter of the transformer.  The default gradient scope covers the weights / biases of the linear
layers inside the transformer blocks (attention projections of both streams, attn2, the MLPs) -- the reference's default
`target_modules` for full fine-tuning and for LoRA; when anything else is trainable (`target_modules: all`: AdaLN modulation
linears, q/k norm weights, timestep / pooled-text MLPs, embedders, proj_out) `denoise_replay` switches the engine to its full
scope for that step (`mi355_engine_set_train_scope`).  `unsupported_reason` reports what cannot be differentiated natively (a
module that is not bound, trainable tensors the engine does not know) and the caller falls back.

Round 4: the same contract for the other families -- `flux_replay` (FLUX.1), `qwen_replay` (Qwen-Image; the true-CFG combine is inside the
engine's forward / backward) and `wan_replay` (Wan; CFG lives in the fused scheduler step, whose adjoint returns d v for both halves of the
forward batch) share ONE autograd node, `_FluxReplayFn`: training-mode transformer forward -> `engine.sde_step` -> `engine.sde_step_bwd` ->
the plan's `backward`.  Gradient buffers are bf16 where the parameter is (`_grad_buffers`: the engines' reduction kernels round on the way out).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .binding import LiveWeights, _LoraWeight, _Plain, _full, unwrap_module


# --------------------------------------------------------------------------------------------- trainable sources
def _lora_trainable(src: _LoraWeight) -> bool:
    lay = src.layer
    return any(lay.lora_A[a].weight.requires_grad or lay.lora_B[a].weight.requires_grad for a in src._active())


def trainable_sources(live: LiveWeights) -> List[Tuple[str, object]]:
    """(engine parameter name, source) for every source that carries a trainable torch tensor."""
    live.sync()                      # resolves the sources against the current module tree
    out = []
    for name, src in live._sources.items():
        if isinstance(src, _Plain):
            if getattr(src.t, "requires_grad", False):
                out.append((name, src))
        elif isinstance(src, _LoraWeight):
            if _lora_trainable(src) or src.layer.base_layer.weight.requires_grad:
                out.append((name, src))
    return out


def _covered_tensor_ids(pairs: Sequence[Tuple[str, object]]) -> set:
    ids = set()
    for _, src in pairs:
        if isinstance(src, _Plain):
            ids.add(id(src.t))
        else:
            lay = src.layer
            ids.add(id(lay.base_layer.weight))
            for a in src._active():
                ids.add(id(lay.lora_A[a].weight))
                ids.add(id(lay.lora_B[a].weight))
    return ids


def unsupported_reason(host) -> Optional[str]:
    """None when the engine's backward covers every trainable parameter of the bound module; else why not."""
    live: Optional[LiveWeights] = getattr(host, "_live_weights", None)
    if live is None:
        return "no torch module is bound to the engine (weights were bound from a state dict)"
    pairs = trainable_sources(live)
    if not pairs:
        return "the bound module has no trainable parameter"
    eng = host.engine
    if not hasattr(eng, "grad_supported"):
        return f"the {type(eng).__name__} has no native backward"
    for name, _ in pairs:
        if not eng.grad_supported(name):
            return f"parameter '{name}' is outside the native backward's scope"
    covered = _covered_tensor_ids(pairs)
    root = unwrap_module(live.get_module())
    for pname, prm in root.named_parameters():
        if prm.requires_grad and id(prm) not in covered:
            return f"trainable parameter '{pname}' has no counterpart in the engine's backward"
    return None


grad_forward_supported = unsupported_reason      # name used by the Flow-Factory plugin


# --------------------------------------------------------------------------------------------- FSDP2 (sharded DTensor parameters)
def _is_sharded(t) -> bool:
    return hasattr(t, "full_tensor") and hasattr(t, "device_mesh")


class _ShardGradBridge(torch.autograd.Function):
    """FSDP2 parameter (a sharded DTensor) -> a stand-in of its GLOBAL shape for the engine's autograd node, which needs the parameter as a
    graph edge only (the values are the engine's own bound copy): no all-gather in the forward.  Backward: the engine hands back the WHOLE
    gradient of this rank's micro-batch; declared a partial value and redistributed to the parameter's placements it is reduce-scattered
    with averaging over the data-parallel mesh -- what FSDP2's own post-backward does for gradients computed by `module.forward`, which the
    engine bypasses (so FSDP2's hooks never fire and nothing is reduced twice)."""

    @staticmethod
    def forward(ctx, p):
        ctx.mesh, ctx.placements = p.device_mesh, p.placements
        return p._local_tensor.new_zeros(()).expand(p.shape)

    @staticmethod
    def backward(ctx, g):
        from torch.distributed.tensor import DTensor, Partial
        if g is None:
            return None
        part = DTensor.from_local(g.contiguous(), ctx.mesh, [Partial("avg")] * ctx.mesh.ndim, run_check=False)
        return part.redistribute(ctx.mesh, ctx.placements)


def _full_with_grad(t: torch.Tensor) -> torch.Tensor:
    """Whole tensor WITH autograd (LoRA factors, whose values enter the merged weight): for an FSDP2 shard the backward averages the
    per-rank gradients over the mesh (`full_tensor()`'s default backward would just chunk this rank's gradient)."""
    if _is_sharded(t) and t.requires_grad:
        from torch.distributed.tensor import Partial
        return t.full_tensor(grad_placements=[Partial("avg")] * t.device_mesh.ndim)
    return _full(t)


def _materialise_with_grad(src, lora_scale: float) -> torch.Tensor:
    """The tensor the engine binds for this source, connected to the trainable leaves by autograd."""
    if isinstance(src, _Plain):
        return _ShardGradBridge.apply(src.t) if _is_sharded(src.t) else src.t
    lay = src.layer
    w = _full_with_grad(lay.base_layer.weight)
    out = w.float() if w.requires_grad else w.detach().float()
    for a in src._active():
        A, B = _full_with_grad(lay.lora_A[a].weight).float(), _full_with_grad(lay.lora_B[a].weight).float()
        delta = (B @ A) * (float(lay.scaling[a]) * float(lora_scale))
        out = out + (delta.t() if getattr(lay, "fan_in_fan_out", False) else delta)
    return out


# --------------------------------------------------------------------------------------------- DDP reducer arming
def _ddp_of(module):
    m = module
    for _ in range(4):
        if type(m).__name__ == "DistributedDataParallel":
            return m
        nxt = getattr(m, "_orig_mod", None)
        if nxt is None:
            return None
        m = nxt
    return None


def _ddp_reducer_armed(host) -> bool:
    """True when a DDP reducer WAITS for this backward's gradients: a DDP wrapper exists and gradient sync is on -- the condition (beside
    grad mode, which is what brought the caller to this autograd node: inside `Function.forward` it reads as off) under which
    `DDP._post_forward` calls `reducer.prepare_for_backward`.  Under `no_sync()` / `accelerator.accumulate`
    (`require_backward_grad_sync` False) a backward without a gradient must leave `param.grad` None, like torch does: dense zeros would
    turn a skipped parameter into a decayed one (AdamW) and cost its memory."""
    lw = getattr(host, "_live_weights", None)
    if lw is None:
        return False
    ddp = _ddp_of(lw.get_module())
    return ddp is not None and bool(getattr(ddp, "require_backward_grad_sync", True))


def _arm_ddp(ddp) -> None:
    """`DDP._pre_forward` as DDP.forward calls it.  With `device_ids` set -- what `accelerator.prepare` does on a GPU -- it moves its
    inputs to the device and indexes the result, so it needs at least one input: an empty tensor on that device (found by the
    world-size-1 RCCL test on the GPU; the gloo tests wrap without device_ids)."""
    ids = getattr(ddp, "device_ids", None)
    if ids:
        ddp._pre_forward(torch.empty(0, device=torch.device(getattr(ddp, "device_type", "cuda"), ids[0])))
    else:
        ddp._pre_forward()


# --------------------------------------------------------------------------------------------- gradient buffers
def _grad_buffers(eng, names, w_meta, device) -> Dict[str, torch.Tensor]:
    """One buffer per trainable engine parameter, registered with the engine.  A bf16 parameter inside the default gradient scope (the blocks'
    linear layers) gets a bf16 buffer the engine writes directly (its reduction kernels round the fp32 sums on the way out: the values of
    `fp32 -> .to(bf16)` without the 4 + 4 + 2 bytes per parameter of HBM round trip); anything else an fp32 buffer, converted by the caller."""
    bufs: Dict[str, torch.Tensor] = {}
    eng.clear_grads()
    # One zero-filled slab per dtype, carved into 256-byte-aligned views (round 6): the reference's default target set is 382 tensors, and
    # 382 `torch.zeros` were 382 fill launches (~0.9 ms of a 87 ms optimize() step, launch-bound) -- now two.  The views are ordinary
    # gradient tensors (`param.grad = view`); the slab lives as long as any of them.
    plan_, totals = [], {torch.bfloat16: 0, torch.float32: 0}
    for name, (shape, dt) in zip(names, w_meta):
        direct = dt == torch.bfloat16 and eng.grad_supported(name) == 1
        bdt = torch.bfloat16 if direct else torch.float32
        numel = 1
        for d in shape:
            numel *= int(d)
        per = 256 // (2 if direct else 4)                       # elements per 256 bytes
        off = totals[bdt]
        totals[bdt] = off + (numel + per - 1) // per * per
        plan_.append((name, tuple(shape), bdt, off, numel))
    slabs = {bdt: torch.zeros(n, device=device, dtype=bdt) for bdt, n in totals.items() if n > 0}
    for name, shape, bdt, off, numel in plan_:
        bufs[name] = slabs[bdt][off:off + numel].view(shape)
        eng.set_grad(name, bufs[name])
    return bufs


def _null_weight_grads(ctx):
    """Weight gradients of a backward that received no gradient at all: None -- except under an ARMED DDP reducer (`_post_forward` ran
    `prepare_for_backward` on these parameters): it must see every parameter it expects, so zeros (a reducer left waiting fails
    the NEXT iteration with "expected to have finished reduction")."""
    if not ctx.ddp_armed:
        return (None,) * len(ctx.names)
    return tuple(torch.zeros(shape, dtype=dt, device=ctx.w_dev) for shape, dt in ctx.w_meta)


# --------------------------------------------------------------------------------------------- the autograd node
class _DenoiseReplayFn(torch.autograd.Function):
    """(log_prob [B], noise_pred [B,C,h,w], next_latents_mean [B,C,h,w]) = step(weights); d/d weights by the engine."""

    @staticmethod
    def forward(ctx, host, plan, names, call, *weights):
        call["_keep"] = {}
        o = plan.denoise_step_train(**call)
        ctx.set_materialize_grads(False)
        ctx.host, ctx.plan, ctx.names, ctx.call = host, plan, names, call
        ctx.w_meta = [(w.shape, w.dtype) for w in weights]
        ctx.w_dev = weights[0].device if weights else None
        ctx.ddp_armed = _ddp_reducer_armed(host)
        ctx.mark_non_differentiable(o.std_dev_t, o.dt)
        return o.log_prob, o.noise_pred, o.next_latents_mean, o.std_dev_t, o.dt

    @staticmethod
    def backward(ctx, g_lp, g_np, g_mean, _g_std, _g_dt):
        host, plan = ctx.host, ctx.plan
        # the weights may have been swapped after the forward (KL reference pass under use_ref_parameters, trainers/grpo.py:281-292):
        # re-bind the current (policy) weights before the data-gradient GEMMs read them
        sync = getattr(host, "_sync_weights", None)
        if sync is not None:
            sync()
        eng = plan.engine
        if g_lp is None and g_np is None and g_mean is None:
            return (None,) * 4 + _null_weight_grads(ctx)
        dev = next(g for g in (g_lp, g_np, g_mean) if g is not None).device
        grads = _grad_buffers(eng, ctx.names, ctx.w_meta, dev)
        try:
            plan.denoise_step_backward(ctx.call, g_lp, g_np, g_mean)
        finally:
            eng.clear_grads()          # (also when the backward raises: no dtype mark may outlive the torch buffer it names)
        outs = tuple(grads[n].to(dt) for n, (_, dt) in zip(ctx.names, ctx.w_meta))
        return (None, None, None, None) + outs


def denoise_replay(host, plan, call: dict):
    """Run the differentiable replay step; returns (log_prob, noise_pred, next_latents_mean, std_dev_t, dt) with autograd attached to
    the trainable parameters behind `host._live_weights`."""
    live: LiveWeights = host._live_weights
    pairs = trainable_sources(live)
    names = [n for n, _ in pairs]
    weights = [_materialise_with_grad(s, live.lora_scale) for _, s in pairs]
    # parameters outside the blocks' linear layers (target_modules: all) need the full gradient scope: more is stashed by the forward
    plan.engine.set_train_scope(any(plan.engine.grad_supported(n) == 2 for n in names))
    ddp = _ddp_of(live.get_module())
    if ddp is not None:
        _arm_ddp(ddp)                           # arms buffer sync / lazy init exactly like DDP.forward
    out = _DenoiseReplayFn.apply(host, plan, names, call, *weights)
    if ddp is not None:
        ddp._post_forward(out[0])               # reducer.prepare_for_backward: bucketed gradient all-reduce during our backward
    return out



# --------------------------------------------------------------------------------------------- FLUX.1 (mi355_flux_forward_train / _backward)
def _train_args(call: dict):
    """Positional arguments of `plan.forward_train` for this step: `call["train_args"]` (Qwen-Image), else the FLUX.1 five."""
    return call.get("train_args") or (call["latents"], call["tm"], call["gm"], call["prompt_embeds"], call["pooled"])


def _cfg_halves(v: torch.Tensor, call: dict):
    """(v_text, v_uncond, guidance) of the fused scheduler step: the network output itself, or -- Wan with classifier-free guidance, whose
    two branches are one forward batch [negative | positive] -- its halves and the guidance scale (`call["cfg_guidance"]`)."""
    g = call.get("cfg_guidance")
    if g is None:
        return v, None, 1.0
    B = v.shape[0] // 2
    return v[B:], v[:B], float(g)


class _FluxReplayFn(torch.autograd.Function):
    """(log_prob [B], noise_pred [B,Ni,C], next_latents_mean [B,Ni,C]) = step(transformer(weights)); d/d weights by the FLUX engine.

    forward  = `FluxPlan.forward_train` (the no-grad forward's kernel binaries on per-block buffers: the velocity is bit-identical) followed by
               the SAME fused scheduler-step kernel the no-grad `forward()` runs -> the replay log-prob equals the rollout's bit for bit;
    backward = `engine.sde_step_bwd` (adjoint of the step: d loss / d v) -> `FluxPlan.backward` (hand-written HIP: head_dim-128 flash-attention
               backward, RoPE + RMSNorm backward, dgrad / wgrad GEMMs) -> fp32 weight gradients handed to autograd."""

    @staticmethod
    def forward(ctx, host, plan, names, call, *weights):
        from .engine import sde_step
        v = plan.forward_train(*_train_args(call))
        vt, vu, g = _cfg_halves(v, call)
        o = sde_step(vt, vu, g, call["latents"], call["sigma"], call["sigma_next"], call["eta"], call["sigma_max"], call["dynamics"],
                     noise=None, next_latents=call["next_latents"], compute_log_prob=call["compute_log_prob"],
                     want=("next_latents_mean", "noise_pred", "std_dev_t", "dt"))
        ctx.set_materialize_grads(False)
        ctx.host, ctx.plan, ctx.names, ctx.call, ctx.v = host, plan, names, call, v
        ctx.serial = plan._train_serial
        ctx.w_meta = [(w.shape, w.dtype) for w in weights]
        ctx.w_dev = weights[0].device if weights else None
        ctx.ddp_armed = _ddp_reducer_armed(host)
        lp = o.log_prob if o.log_prob is not None else torch.zeros((vt.shape[0],), device=v.device)
        ctx.mark_non_differentiable(o.std_dev_t, o.dt)
        return lp, o.noise_pred, o.next_latents_mean, o.std_dev_t, o.dt

    @staticmethod
    def backward(ctx, g_lp, g_np, g_mean, _g_std, _g_dt):
        from .engine import sde_step_bwd
        host, plan, call = ctx.host, ctx.plan, ctx.call
        sync = getattr(host, "_sync_weights", None)      # the weights may have been swapped after the forward (KL reference pass): re-bind
        if sync is not None:
            sync()
        if g_lp is None and g_np is None and g_mean is None:
            return (None,) * 4 + _null_weight_grads(ctx)
        if ctx.serial != plan._train_serial:
            # another training forward ran on this plan since (a loss that sums several grad forwards before one backward): the stash
            # holds ITS activations -- re-run this step's forward on the kept inputs (same kernels, same weights: bit-identical stash)
            plan.recomputed_forwards = getattr(plan, "recomputed_forwards", 0) + 1
            plan.forward_train(*_train_args(call))
        if not call["compute_log_prob"]:
            g_lp = None
        vt, vu, g = _cfg_halves(ctx.v, call)
        dv = sde_step_bwd(vt, vu, g, call["latents"], call["next_latents"], call["sigma"], call["sigma_next"], call["eta"],
                          call["sigma_max"], call["dynamics"], call["compute_log_prob"], g_lp, g_np, g_mean)      # [uncond | text] with CFG
        eng = plan.engine
        grads = _grad_buffers(eng, ctx.names, ctx.w_meta, dv.device)
        try:
            plan.backward(dv)
        finally:
            eng.clear_grads()          # (also when the backward raises: no dtype mark may outlive the torch buffer it names)
        outs = tuple(grads[n].to(dt) for n, (_, dt) in zip(ctx.names, ctx.w_meta))
        return (None, None, None, None) + outs


def flux_replay(host, plan, call: dict):
    """The differentiable FLUX.1 replay step; returns (log_prob, noise_pred, next_latents_mean, std_dev_t, dt) with autograd attached to
    the trainable parameters behind `host._live_weights` (LoRA-merged weights, FSDP2 shards and DDP reducer arming as `denoise_replay`)."""
    live: LiveWeights = host._live_weights
    pairs = trainable_sources(live)
    names = [n for n, _ in pairs]
    weights = [_materialise_with_grad(s, live.lora_scale) for _, s in pairs]
    ddp = _ddp_of(live.get_module())
    if ddp is not None:
        _arm_ddp(ddp)
    out = _FluxReplayFn.apply(host, plan, names, call, *weights)
    if ddp is not None:
        ddp._post_forward(out[0])
    return out


def qwen_replay(host, plan, call: dict):
    """The differentiable Qwen-Image replay step (mi355_qwen_forward_train / _backward: true-CFG combine included): the FLUX.1 node with
    `call["train_args"] = (latents, t_model, embeds, lens, guidance_scale)`."""
    assert "train_args" in call
    return flux_replay(host, plan, call)



def wan_replay(host, plan, call: dict):
    """The differentiable Wan replay step (mi355_wan_forward_train / _backward; CFG = `u + g (c - u)` inside the fused scheduler step, whose
    adjoint returns d v for both halves): the FLUX.1 node with `call["train_args"] = (latents, t, enc_a, enc_b)` and `call["cfg_guidance"]`."""
    assert "train_args" in call and "cfg_guidance" in call
    return flux_replay(host, plan, call)
