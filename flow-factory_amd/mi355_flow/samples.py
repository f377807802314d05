"""Result container of a rollout: field-for-field the subset of the reference's `SD3_5Sample` /
`BaseSample` (reference src/flow_factory/models/stable_diffusion/sd3_5.py:50-58,
src/flow_factory/samples/samples.py:68-107) that the GRPO trainer reads.  Tensors carry no batch
dimension; `latent_index_map` / `log_prob_index_map` are shared across a batch."""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch


@dataclass
class SD3_5Sample:
    # denoising trajectory
    timesteps: Optional[torch.Tensor] = None            # (N,)
    all_latents: Optional[torch.Tensor] = None          # (P, C, h, w) storage dtype, kept positions only
    latent_index_map: Optional[torch.Tensor] = None     # (N+1,) position -> row of all_latents, -1 = dropped
    log_probs: Optional[torch.Tensor] = None            # (P',) fp32, trained (SDE) steps only
    log_prob_index_map: Optional[torch.Tensor] = None   # (N+1,)
    # output dimensions / media
    height: Optional[int] = None
    width: Optional[int] = None
    image: Optional[torch.Tensor] = None                # (3, H, W) in [0, 1] when a VAE decoder is attached
    # prompt
    prompt: Optional[str] = None
    prompt_ids: Optional[torch.Tensor] = None
    prompt_embeds: Optional[torch.Tensor] = None
    pooled_prompt_embeds: Optional[torch.Tensor] = None
    negative_prompt: Optional[str] = None
    negative_prompt_ids: Optional[torch.Tensor] = None
    negative_prompt_embeds: Optional[torch.Tensor] = None
    negative_pooled_prompt_embeds: Optional[torch.Tensor] = None
    extra_kwargs: Dict[str, Any] = field(default_factory=dict)
    _unique_id: Optional[int] = field(default=None, repr=False, compare=False)

    @property
    def unique_id(self) -> int:
        """Stable 63-bit id of the prompt identity (groups the K repeats of one prompt)."""
        if self._unique_id is None:
            h = hashlib.sha256()
            for v in (self.prompt, self.prompt_ids, self.negative_prompt, self.negative_prompt_ids):
                if v is None:
                    h.update(b"\x00")
                elif isinstance(v, torch.Tensor):
                    h.update(v.detach().cpu().contiguous().numpy().tobytes())
                else:
                    h.update(str(v).encode())
            self._unique_id = int.from_bytes(h.digest()[:8], "big") >> 1
        return self._unique_id

    def to(self, device) -> "SD3_5Sample":
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        self.extra_kwargs = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.extra_kwargs.items()}
        return self


@dataclass
class Flux1Sample(SD3_5Sample):
    """`Flux1Sample` (reference src/flow_factory/models/flux/flux1.py:53-60): packed latents `(P, Ni, 64)` in `all_latents`,
    plus the `img_ids` shared by the batch.  No negative prompt (guidance is embedded)."""
    img_ids: Optional[torch.Tensor] = None


@dataclass
class WanT2VSample(SD3_5Sample):
    """`WanT2VSample` (reference src/flow_factory/models/wan/wan2_t2v.py): video latents `(P, 16, T, h, w)` in `all_latents`, the decoded
    clip in `video` (None unless a video decoder is attached)."""
    video: Optional[torch.Tensor] = None
