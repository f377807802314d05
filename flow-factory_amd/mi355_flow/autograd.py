"""Differentiable engine forward for the `optimize()` replay (reference trainers/grpo.py:185-342).

Placeholder until the backward kernels land in this round: `grad_forward_supported` reports why the native path is not
taken, and the plugin then uses the reference's autograd path."""
from __future__ import annotations

from typing import Optional


def grad_forward_supported(adapter) -> Optional[str]:
    return "the native backward is not built yet"


def sd3_grad_forward(adapter, *args, **kwargs):
    raise NotImplementedError("mi355_flow: native grad-mode forward is not built yet")
