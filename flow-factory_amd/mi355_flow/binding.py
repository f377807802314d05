"""Keeps an engine's packed weights equal to the CURRENT parameters of a (possibly wrapped) torch module.

GRPO weights are live: `optimizer.step()`, EMA swaps (`use_ema_parameters`), reference swaps (`use_ref_parameters`,
`use_named_parameters`), LoRA adapters being toggled (`disable_adapter()` for the KL reference, reference
src/flow_factory/models/abc.py:523-597,660-682) all change what `self.transformer(...)` would compute.  The engine holds a
re-packed bf16 COPY (`mi355_*_bind_weight`), so every engine call is preceded by `LiveWeights.sync()`:

  * every engine parameter name is resolved against the module tree (DDP `.module`, torch.compile `._orig_mod` and peft
    `PeftModel.base_model.model` wrappers are peeled; FSDP2 `DTensor` parameters are gathered with `.full_tensor()`);
  * a peft LoRA layer (`base_layer` + `lora_A` / `lora_B` / `scaling`) binds the MERGED weight `W + sum_a s_a * B_a @ A_a`
    (what the adapter-enabled forward computes), or the bare base weight while the adapter is disabled / already merged;
    DoRA and LoRA-bias layers raise (no silent approximation, reference constraints.md:144-145);
  * a tensor is re-bound only when its key changed: (storage pointer, autograd version counter, [adapter state], epoch).
    In-place optimizer updates bump the version counter; `param.data.copy_()` swaps (EMA / ref contexts) do NOT, so the
    owner bumps `epoch` via `invalidate()` whenever such a context is entered or left and on every mode switch -- trainable
    (and LoRA-composite) tensors then re-bind, frozen ones are skipped.

Missing names raise unless `partial=True` (a stale weight must never be used silently).
"""
from __future__ import annotations

import weakref
from typing import Callable, Dict, Iterable, List, Tuple

import torch


def unwrap_module(m):
    """Peel DDP / FSDP1 / torch.compile / peft wrappers down to the module whose attribute paths are the HF names."""
    seen = 0
    while seen < 8:
        seen += 1
        if hasattr(m, "_orig_mod"):                       # torch.compile
            m = m._orig_mod
        elif hasattr(m, "peft_config") and hasattr(m, "base_model"):     # peft.PeftModel -> LoraModel -> model
            inner = m.base_model
            m = getattr(inner, "model", inner)
        elif type(m).__name__ in ("DistributedDataParallel", "FullyShardedDataParallel", "DeepSpeedEngine") and hasattr(m, "module"):
            m = m.module
        else:
            break
    return m


def _deepspeed_step(m):
    """Optimizer-step counter of a DeepSpeedEngine wrapper anywhere in the wrapper chain (None without DeepSpeed).  `global_steps` advances
    at every gradient-accumulation boundary (`DeepSpeedEngine._take_model_step`), i.e. whenever the parameters may have been rewritten."""
    for _ in range(8):
        if type(m).__name__ == "DeepSpeedEngine":
            return (int(getattr(m, "global_steps", 0)), int(getattr(m, "skipped_steps", 0)))
        nxt = getattr(m, "_orig_mod", None) or (getattr(m, "module", None) if type(m).__name__ in ("DistributedDataParallel",) else None)
        if nxt is None:
            return None
        m = nxt
    return None


def _is_lora_layer(mod) -> bool:
    return hasattr(mod, "base_layer") and hasattr(mod, "lora_A") and hasattr(mod, "lora_B")


def _full(t: torch.Tensor) -> torch.Tensor:
    """FSDP2 shards -> the whole tensor (collective: every rank syncs at the same point of the epoch)."""
    return t.full_tensor() if hasattr(t, "full_tensor") else t


def _tkey(t: torch.Tensor) -> Tuple[int, int, int]:
    # FSDP2: an in-place optimizer update of a DTensor parameter bumps the version counter of the DTensor wrapper, not the one of its
    # `_local_tensor` (measured: torch 2.10, fully_shard + SGD.step) -- key on both, and on the local shard's storage pointer
    loc = getattr(t, "_local_tensor", t)
    return (loc.data_ptr(), t._version, loc._version)


class _Plain:
    def __init__(self, tensor: torch.Tensor):
        self.t = tensor

    def key(self, epoch, lora_scale=1.0):
        return (_tkey(self.t), epoch if getattr(self.t, "requires_grad", False) else 0)

    def materialise(self, lora_scale=1.0) -> torch.Tensor:
        return _full(self.t.detach())


class _LoraWeight:
    """`W + sum_a scaling[a] * lora_B[a].weight @ lora_A[a].weight` of a peft LoRA `Linear` (adapter-enabled forward)."""

    def __init__(self, layer, path: str):
        self.layer, self.path = layer, path
        mag = getattr(layer, "lora_magnitude_vector", None)
        if mag is not None and len(mag) > 0:
            raise NotImplementedError(f"mi355_flow: DoRA layer at '{path}' is not supported by the native engine binding")
        if any(getattr(layer, "lora_bias", {}).values()) if isinstance(getattr(layer, "lora_bias", None), dict) else False:
            raise NotImplementedError(f"mi355_flow: LoRA layer at '{path}' carries lora_bias: not supported")

    def _active(self) -> List[str]:
        lay = self.layer
        if bool(getattr(lay, "disable_adapters", False)) or bool(getattr(lay, "merged", False)):
            return []          # disabled: base weight only; merged: the base weight already contains the delta
        act = getattr(lay, "active_adapters", None)
        if act is None:
            act = getattr(lay, "active_adapter", [])
        if isinstance(act, str):
            act = [act]
        return [a for a in act if a in lay.lora_A]

    def key(self, epoch, lora_scale=1.0):
        lay = self.layer
        parts = [_tkey(lay.base_layer.weight), float(lora_scale)]
        for a in self._active():
            parts.append((a, _tkey(lay.lora_A[a].weight), _tkey(lay.lora_B[a].weight), float(lay.scaling[a])))
        return (tuple(parts), bool(getattr(lay, "disable_adapters", False)), bool(getattr(lay, "merged", False)), epoch)

    def materialise(self, lora_scale=1.0) -> torch.Tensor:
        lay = self.layer
        w = _full(lay.base_layer.weight.detach())
        act = self._active()
        if not act:
            return w
        out = w.float()
        for a in act:
            A = _full(lay.lora_A[a].weight.detach()).float()
            B = _full(lay.lora_B[a].weight.detach()).float()
            if getattr(lay, "fan_in_fan_out", False):
                out = out + float(lay.scaling[a]) * float(lora_scale) * (B @ A).t()
            else:
                out = out + float(lay.scaling[a]) * float(lora_scale) * (B @ A)
        return out


def resolve_sources(root, names: Iterable[str], partial: bool = False) -> Dict[str, object]:
    """engine parameter name -> source object, walking attribute paths from the unwrapped root."""
    root = unwrap_module(root)
    out: Dict[str, object] = {}
    missing: List[str] = []
    for name in names:
        parts = name.split(".")
        mod, ok = root, True
        for i, p in enumerate(parts[:-1]):
            if _is_lora_layer(mod):        # path continues below a wrapped layer (never for Linear leaves)
                mod = mod.base_layer
            nxt = getattr(mod, p, None) if not p.isdigit() else (mod[int(p)] if hasattr(mod, "__getitem__") else getattr(mod, p, None))
            if nxt is None:
                ok = False
                break
            mod = nxt
        leaf = parts[-1]
        if ok and _is_lora_layer(mod):
            if leaf == "weight":
                out[name] = _LoraWeight(mod, ".".join(parts[:-1]))
                continue
            mod = mod.base_layer
        t = getattr(mod, leaf, None) if ok else None
        if isinstance(t, torch.Tensor):
            out[name] = _Plain(t)
        else:
            missing.append(name)
    if missing and not partial:
        raise KeyError(f"mi355_flow: the module lacks {len(missing)} of the engine's parameters (first: '{missing[0]}'); "
                       "refusing to run on stale or partially bound weights")
    return out


class LiveWeights:
    """`sync()` before every engine call; `invalidate()` on mode switches and parameter-swap contexts."""

    def __init__(self, engine, get_module: Callable[[], object], partial: bool = False):
        self.engine, self.get_module, self.partial = engine, get_module, partial
        self.epoch = 1
        self.lora_scale = 1.0      # joint_attention_kwargs['scale'] (diffusers scale_lora_layers): extra factor on every LoRA delta
        self._root_ref = None      # weak reference to the unwrapped module the sources were resolved against (an `id()` could be reused
                                   # by a NEW module allocated at a freed one's address: the sources would silently stay on the old tensors)
        self._sources: Dict[str, object] = {}
        self._bound: Dict[str, object] = {}
        self.last_rebinds = 0
        self._ds_step = None       # (global_steps, micro_steps of the last optimizer boundary) of a DeepSpeedEngine wrapper, if any

    def invalidate(self) -> None:
        self.epoch += 1

    def reset(self) -> None:
        """Re-resolve every source at the next `sync()`: the module TREE was changed in place (e.g. linear layers wrapped into LoRA layers
        after the first sync), which neither version counters nor the root's identity can show."""
        self._root_ref = None
        self._bound.clear()

    def set_lora_scale(self, scale: float) -> None:
        self.lora_scale = float(scale)

    @property
    def has_lora(self) -> bool:
        return any(isinstance(s, _LoraWeight) for s in self._sources.values())

    def sync(self) -> int:
        root = self.get_module()
        inner = unwrap_module(root)
        # DeepSpeed ZeRO-1/2 and its BF16_Optimizer write updated parameters through `.data.copy_()` into views of a flat buffer: neither the
        # storage pointer nor the version counter moves (reference config/deepspeed/*.yaml).  Its engine counts optimizer steps: a new count
        # re-binds every trainable tensor.
        step = _deepspeed_step(root)
        if step != self._ds_step:
            self._ds_step = step
            self.invalidate()
        if self._root_ref is None or self._root_ref() is not inner:
            self._sources = resolve_sources(inner, self.engine.param_names(), partial=self.partial)
            self._root_ref = weakref.ref(inner)
            self._bound.clear()
        n = 0
        for name, src in self._sources.items():
            k = src.key(self.epoch, self.lora_scale)
            if self._bound.get(name) == k:
                continue
            self.engine.bind_tensor(name, src.materialise(self.lora_scale))
            self._bound[name] = k
            n += 1
        if n:
            self.engine.finish_binding()
        self.last_rebinds = n
        return n
