"""Result containers of a rollout, API-compatible with the reference's `BaseSample` / `SD3_5Sample` / `Flux1Sample` /
`WanT2VSample` (reference src/flow_factory/samples/samples.py:68-375, models/stable_diffusion/sd3_5.py:50-58,
models/flux/flux1.py:53-59, models/wan/wan2_t2v.py:49-51) for everything the GRPO trainer touches:

  * field set + `extra_kwargs` overflow (`to_dict` flattens it, `sample['key']` / `sample.key` read through it);
  * `BaseSample.stack(samples)` -> batched dict (`optimize()`, trainers/grpo.py:215): shared fields (`height`, `width`,
    `latent_index_map`, `log_prob_index_map`, + per-class extras) take the first element, same-shape tensors stack, dicts
    recurse, the rest becomes lists;
  * `unique_id`: sha256 over prompt (utf-8) or prompt_ids bytes, then the negative prompt likewise, first 8 digest bytes as a
    SIGNED big-endian integer -- the same value the reference computes, so group identity matches across the two
    implementations (advantage grouping, `collect_group_rewards`);
  * `image` canonicalised to a `(C, H, W)` tensor on construction (`__post_init__`, samples.py:141-146).

Under a Flow-Factory installation the plugin returns the REFERENCE's own classes (`_sample_cls` hook of the rollout
mixins); these mirrors serve the standalone adapters.  Tensors carry no batch dimension.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field, fields
from typing import Any, ClassVar, Dict, List, Optional, Union

import numpy as np
import torch


def _image_to_chw(img) -> torch.Tensor:
    """One image in any of the reference's accepted forms -> float/uint8 tensor (C, H, W) (utils/base.py
    `standardize_image_batch(..., 'pt')[0]`): tensors keep dtype/range; HWC numpy arrays and PIL images become CHW."""
    if isinstance(img, torch.Tensor):
        t = img
        if t.dim() == 4 and t.shape[0] == 1:
            t = t[0]
        if t.dim() == 2:
            t = t.unsqueeze(0)
        if t.dim() != 3:
            raise ValueError(f"image tensor must be (C,H,W) or (1,C,H,W), got {tuple(img.shape)}")
        if t.shape[0] not in (1, 3, 4) and t.shape[-1] in (1, 3, 4):   # HWC tensor
            t = t.permute(2, 0, 1)
        return t
    if isinstance(img, np.ndarray):
        a = img[0] if img.ndim == 4 and img.shape[0] == 1 else img
        if a.ndim == 2:
            a = a[..., None]
        t = torch.from_numpy(np.ascontiguousarray(a))
        if t.shape[-1] in (1, 3, 4):
            t = t.permute(2, 0, 1)
        if t.dtype == torch.uint8:
            t = t.float() / 255.0
        return t
    if hasattr(img, "convert") and hasattr(img, "size"):           # PIL.Image without importing PIL
        a = np.asarray(img.convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(a.copy()).permute(2, 0, 1).float() / 255.0
    raise TypeError(f"unsupported image type {type(img).__name__}")


def _is_tensor_list(v) -> bool:
    return isinstance(v, (list, tuple)) and len(v) > 0 and any(isinstance(t, torch.Tensor) for t in v)


@dataclass
class BaseSample:
    """One rollout sample (reference samples.py:68-107)."""
    _id_fields: ClassVar[frozenset] = frozenset({"prompt", "prompt_ids", "negative_prompt", "negative_prompt_ids"})
    _shared_fields: ClassVar[frozenset] = frozenset({"height", "width", "latent_index_map", "log_prob_index_map"})

    # denoising trajectory
    timesteps: Optional[torch.Tensor] = None            # (N,)
    all_latents: Optional[torch.Tensor] = None          # (P, ...) storage dtype, kept positions only
    latent_index_map: Optional[torch.Tensor] = None     # (N+1,) position -> row of all_latents, -1 = dropped
    log_probs: Optional[torch.Tensor] = None            # (P',) fp32, trained (SDE) steps only
    log_prob_index_map: Optional[torch.Tensor] = None   # (N+1,)
    # output dimensions / media
    height: Optional[int] = None
    width: Optional[int] = None
    image: Optional[Any] = None                         # -> (C, H, W) tensor
    video: Optional[Any] = None                         # (T, C, H, W) tensor when a video decoder is attached
    audio: Optional[torch.Tensor] = None
    audio_sample_rate: Optional[int] = None
    # prompt
    prompt: Optional[str] = None
    prompt_ids: Optional[torch.Tensor] = None
    prompt_embeds: Optional[torch.Tensor] = None
    negative_prompt: Optional[str] = None
    negative_prompt_ids: Optional[torch.Tensor] = None
    negative_prompt_embeds: Optional[torch.Tensor] = None
    extra_kwargs: Dict[str, Any] = field(default_factory=dict)
    _unique_id: Optional[int] = field(default=None, repr=False, compare=False)

    def __post_init__(self):
        if self.image is not None:
            self.image = _image_to_chw(self.image)
        if self.video is not None and not isinstance(self.video, torch.Tensor):
            self.video = torch.stack([_image_to_chw(f) for f in self.video])
        if isinstance(self.audio, torch.Tensor) and self.audio.dim() == 1:
            self.audio = self.audio.unsqueeze(0)

    # ------------------------------------------------------------------ mapping protocol
    @classmethod
    def shared_fields(cls) -> frozenset:
        out = set()
        for klass in cls.__mro__:
            out |= set(vars(klass).get("_shared_fields", ()))
        return frozenset(out)

    def to_dict(self) -> Dict[str, Any]:
        d = {f.name: getattr(self, f.name) for f in fields(self) if f.name != "extra_kwargs"}
        d.update(self.extra_kwargs)
        return d

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "BaseSample":
        names = {f.name for f in fields(cls)}
        extra = dict(d.get("extra_kwargs") or {})
        clash = set(extra) & (names - {"extra_kwargs"})
        if clash:
            raise ValueError(f"extra_kwargs contains reserved field names: {clash}")
        extra.update({k: v for k, v in d.items() if k not in names})
        return cls(**{k: v for k, v in d.items() if k in names and k != "extra_kwargs"}, extra_kwargs=extra)

    def __getattr__(self, key: str) -> Any:      # only reached when normal lookup fails
        extra = self.__dict__.get("extra_kwargs")
        if extra is not None and key in extra:
            return extra[key]
        raise AttributeError(f"'{type(self).__name__}' has no attribute '{key}'")

    def __setattr__(self, key: str, value: Any) -> None:
        if key in type(self)._id_fields:
            object.__setattr__(self, "_unique_id", None)
        object.__setattr__(self, key, value)

    def keys(self):
        return self.to_dict().keys()

    def __getitem__(self, key: str) -> Any:
        try:
            return getattr(self, key)
        except AttributeError:
            raise KeyError(f"Key '{key}' not found in {type(self).__name__}") from None

    def __iter__(self):
        return iter(self.keys())

    def short_rep(self) -> Dict[str, Any]:
        return {k: (f"Tensor{tuple(v.shape)}" if isinstance(v, torch.Tensor) and v.numel() > 16 else v)
                for k, v in self.to_dict().items()}

    def to(self, device: Union[torch.device, str], depth: int = 1) -> "BaseSample":
        assert 0 <= depth <= 1, "Only depth 0 and 1 are supported."
        device = torch.device(device)
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, torch.Tensor):
                setattr(self, f.name, v.to(device))
            elif depth == 1 and _is_tensor_list(v):
                setattr(self, f.name, [t.to(device) if isinstance(t, torch.Tensor) else t for t in v])
        return self

    # ------------------------------------------------------------------ group identity
    def _hash_id_fields(self, hasher) -> None:
        for text, ids in ((self.prompt, self.prompt_ids), (self.negative_prompt, self.negative_prompt_ids)):
            if text is not None:
                hasher.update(text.encode("utf-8"))
            elif ids is not None:
                hasher.update(ids.cpu().numpy().tobytes())

    def compute_unique_id(self, num_bytes: int = 8) -> int:
        if not 1 <= num_bytes <= 32:
            raise ValueError(f"num_bytes must be in [1, 32] (sha256 digest), got {num_bytes}")
        hasher = hashlib.sha256()
        self._hash_id_fields(hasher)
        return int.from_bytes(hasher.digest()[:num_bytes], byteorder="big", signed=True)

    @property
    def unique_id(self) -> int:
        if self._unique_id is None:
            object.__setattr__(self, "_unique_id", self.compute_unique_id())
        return self._unique_id

    def reset_unique_id(self) -> None:
        object.__setattr__(self, "_unique_id", None)

    # ------------------------------------------------------------------ batching (optimize(): trainers/grpo.py:215)
    @classmethod
    def _stack_values(cls, key: str, values: List[Any]):
        if not values:
            return values
        if all(v is None for v in values):
            return None
        head = values[0]
        if key in cls.shared_fields():
            return head
        if isinstance(head, torch.Tensor):
            return torch.stack(values) if all(v.shape == head.shape for v in values) else values
        if isinstance(head, dict) and all(isinstance(v, dict) for v in values):
            return {k: cls._stack_values(k, [v[k] for v in values]) for k in head}
        return values

    @classmethod
    def stack(cls, samples: List["BaseSample"]) -> Dict[str, Any]:
        if not samples:
            raise ValueError("No samples to stack.")
        kind = type(samples[0])
        rows = [s.to_dict() for s in samples]
        return {k: kind._stack_values(k, [r[k] for r in rows]) for k in rows[0]}


@dataclass
class SD3_5Sample(BaseSample):
    """reference sd3_5.py:50-58"""
    _shared_fields: ClassVar[frozenset] = frozenset()
    pooled_prompt_embeds: Optional[torch.Tensor] = None
    negative_pooled_prompt_embeds: Optional[torch.Tensor] = None


@dataclass
class Flux1Sample(BaseSample):
    """reference flux1.py:53-59: packed latents `(P, Ni, 64)` in `all_latents`, `img_ids` shared by the batch; no negative
    prompt (guidance is embedded)."""
    _shared_fields: ClassVar[frozenset] = frozenset({"img_ids"})
    pooled_prompt_embeds: Optional[torch.Tensor] = None
    img_ids: Optional[torch.Tensor] = None


@dataclass
class QwenImageSample(BaseSample):
    """reference qwen_image.py:54-62: packed latents `(P, Ni, 64)` in `all_latents`; per-sample text masks (ragged prompts stay
    ragged until `_pad_batch_prompt`), `img_shapes = [(1, h/16, w/16)]`."""
    _shared_fields: ClassVar[frozenset] = frozenset()
    prompt_embeds_mask: Optional[torch.Tensor] = None
    negative_prompt_embeds_mask: Optional[torch.Tensor] = None
    img_shapes: Optional[List[Any]] = None


@dataclass
class WanT2VSample(BaseSample):
    """reference wan2_t2v.py:49-51: video latents `(P, 16, T, h, w)` in `all_latents`, the decoded clip in `video`."""
    _shared_fields: ClassVar[frozenset] = frozenset()
