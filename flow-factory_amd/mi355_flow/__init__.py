"""mi355_flow -- MI355X-native GRPO rollout engine for SD3.5-medium behind Flow-Factory's adapter API.

Host side (this package) mirrors the reference's operator interface for the rollout hot path;
the arithmetic lives in libmi355flow.so (hand-written HIP for gfx950, C ABI in include/mi355_flow.h).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
