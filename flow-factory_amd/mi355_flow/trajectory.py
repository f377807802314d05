"""Sparse trajectory bookkeeping of a rollout (host side).

Mirrors the public behaviour of the reference's `TrajectoryCollector`, `CallbackCollector` and
`compute_trajectory_indices` (reference src/flow_factory/utils/trajectory_collector.py:40-180,
:187-337, :344-388): which trajectory positions are kept, the compact storage order and the dense
`position -> compact index` maps (-1 = not kept) that `optimize()` uses to find x_i / x_{i+1}.
The engine consumes the same selection as a `keep_slot` table (see `keep_slots`).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Any, Dict, List, Literal, Optional, Set, Union

import torch

TrajectoryIndicesType = Union[Literal["all"], List[int], None]


def _resolve(indices: TrajectoryIndicesType, n_positions: int) -> Optional[Set[int]]:
    """None -> keep everything; empty set -> keep nothing; else the normalised positions."""
    if indices is None:
        return set()
    if isinstance(indices, str):
        if indices != "all":
            raise ValueError(f"trajectory indices must be 'all', None or a list of ints, got {indices!r}")
        return None
    keep = set()
    for i in indices:
        i = int(i)
        if i < 0:
            i += n_positions
        if 0 <= i < n_positions:
            keep.add(i)
    return keep


class TrajectoryCollector:
    """Keeps tensors at selected trajectory positions 0..T (T = total_steps)."""

    def __init__(self, indices: TrajectoryIndicesType = "all", total_steps: int = 0):
        self.indices = indices
        self.total_steps = total_steps
        self._keep = _resolve(indices, total_steps + 1)
        self._values: List[torch.Tensor] = []
        self._positions: List[int] = []

    @property
    def is_disabled(self) -> bool:
        return self._keep is not None and not self._keep

    @property
    def collect_all(self) -> bool:
        return self._keep is None

    def should_collect(self, step_idx: int) -> bool:
        return self._keep is None or step_idx in self._keep

    def collect(self, value: torch.Tensor, step_idx: int) -> None:
        if self.should_collect(step_idx):
            self._values.append(value)
            self._positions.append(step_idx)

    def get_result(self) -> Optional[List[torch.Tensor]]:
        return None if self.is_disabled else self._values

    @property
    def collected_indices(self) -> List[int]:
        return self._positions

    def get_index_map(self) -> Optional[torch.Tensor]:
        if self.is_disabled:
            return None
        n = self.total_steps + 1
        if self.collect_all:
            return torch.arange(n, dtype=torch.long)
        out = torch.full((n,), -1, dtype=torch.long)
        for slot, pos in enumerate(self._positions):
            out[pos] = slot
        return out

    def reset(self) -> None:
        self._values, self._positions = [], []

    def __len__(self) -> int:
        return len(self._values)


class CallbackCollector:
    """Collects named per-step values (step indices 0..T-1) with the same gating."""

    def __init__(self, indices: TrajectoryIndicesType = "all", total_steps: int = 0):
        self._gate = TrajectoryCollector(indices=indices, total_steps=total_steps)
        self._data: Dict[str, List] = defaultdict(list)
        self._steps: List[int] = []

    @property
    def is_disabled(self) -> bool:
        return self._gate.is_disabled

    def should_collect(self, step_idx: int) -> bool:
        return self._gate.should_collect(step_idx)

    def collect_step(self, step_idx: int, output: Any, keys: List[str], capturable: Optional[Dict[str, Any]] = None) -> None:
        if not keys or not self.should_collect(step_idx):
            return
        if step_idx not in self._steps:
            self._steps.append(step_idx)
        for key in keys:
            val = None
            if capturable and capturable.get(key) is not None:
                val = capturable[key]
            elif hasattr(output, key):
                val = getattr(output, key)
            if val is not None:
                self._data[key].append(val)

    def get_result(self) -> Dict[str, Any]:
        out = {}
        for k, v in self._data.items():
            out[k] = torch.stack(v, dim=1) if v and isinstance(v[0], torch.Tensor) else v
        return out

    def get_index_map(self) -> Optional[torch.Tensor]:
        if self.is_disabled:
            return None
        T = self._gate.total_steps
        if self._gate.collect_all:
            return torch.arange(T, dtype=torch.long)
        out = torch.full((T,), -1, dtype=torch.long)
        for slot, step in enumerate(self._steps):
            if 0 <= step < T:
                out[step] = slot
        return out

    @property
    def collected_indices(self) -> List[int]:
        return self._steps

    def reset(self) -> None:
        self._data = defaultdict(list)
        self._steps = []

    def __len__(self) -> int:
        return len(self._steps)


def compute_trajectory_indices(train_timestep_indices, num_inference_steps: int, include_initial: bool = False) -> List[int]:
    """Positions {i, i+1 : i in train steps} (deduplicated, sorted): what GRPO's replay needs."""
    if isinstance(train_timestep_indices, torch.Tensor):
        train_timestep_indices = train_timestep_indices.tolist()
    n = num_inference_steps + 1
    pos = {0} if include_initial else set()
    for i in train_timestep_indices:
        for j in (i, i + 1):
            if 0 <= j < n:
                pos.add(int(j))
    return sorted(pos)


def create_trajectory_collector(indices: TrajectoryIndicesType, num_steps: int) -> TrajectoryCollector:
    return TrajectoryCollector(indices=indices, total_steps=num_steps)


def create_callback_collector(indices: TrajectoryIndicesType, num_steps: int) -> CallbackCollector:
    return CallbackCollector(indices=indices, total_steps=num_steps)


def keep_slots(indices: TrajectoryIndicesType, num_steps: int) -> List[int]:
    """`position -> output slot` table (N+1 entries, -1 = dropped) handed to mi355_rollout; equals
    TrajectoryCollector(indices, N).get_index_map() for a full rollout."""
    keep = _resolve(indices, num_steps + 1)
    if keep is None:
        return list(range(num_steps + 1))
    out, slot = [], 0
    for pos in range(num_steps + 1):
        if pos in keep:
            out.append(slot)
            slot += 1
        else:
            out.append(-1)
    return out


class CollectedRollout:
    """What every adapter's `inference()` hands to its sample class after a rollout: the reference's collector bookkeeping
    (models/stable_diffusion/sd3_5.py:265-304 and its siblings) applied to the engine's outputs."""

    __slots__ = ("latents", "log_probs", "latent_index_map", "log_prob_index_map", "callbacks", "callback_index_map")

    def per_sample(self, b: int) -> Dict[str, Any]:
        """The trajectory keyword arguments of sample `b` (latents (P, ...), log-probs (P',), the shared index maps, callback tensors)."""
        return dict(all_latents=self.latents[b] if self.latents is not None else None,
                    log_probs=self.log_probs[b] if self.log_probs is not None else None,
                    latent_index_map=self.latent_index_map, log_prob_index_map=self.log_prob_index_map,
                    extra_kwargs={**{k: v[b] for k, v in self.callbacks.items()}, "callback_index_map": self.callback_index_map})


def collect_rollout(trajectory_indices: TrajectoryIndicesType, num_steps: int, latent_at, log_probs, noise_levels, compute_log_prob: bool,
                    step_outputs=None, extra_keys=(), captured_noise_levels=None, dynamics: Optional[str] = None) -> CollectedRollout:
    """`latent_at(pos)` -> the stored latents (B, ...) at trajectory position pos (only asked for collected positions);
    `log_probs[i]` (B,) for SDE steps; `step_outputs[i]` the per-step scheduler outputs when callback tensors were requested."""
    N = num_steps
    lat_c = create_trajectory_collector(trajectory_indices, N)
    lp_c = create_trajectory_collector(trajectory_indices, N) if compute_log_prob else None
    cb_c = create_callback_collector(trajectory_indices, N)
    if lat_c.should_collect(0):
        lat_c.collect(latent_at(0), 0)
    for i in range(N):
        if lat_c.should_collect(i + 1):
            lat_c.collect(latent_at(i + 1), i + 1)
        # the reference gates on the SCHEDULE's noise level (`get_noise_level_for_timestep(t) > 0`, sd3_5.py:275-278), not on the effective
        # one.  In evaluation mode (effective level 0 on every step) `step()` still evaluates its log-prob expression: under Flow-SDE /
        # Dance-SDE that is -(x' - mean)^2 / (2 * 0) - log(0) = NaN in fp32 (flow_match_euler_discrete.py:363-398), under ODE dynamics
        # `zeros(B)` (:337-340) -- and that value IS collected, so `log_probs` / `log_prob_index_map` exist in the sample.
        sched_level = (captured_noise_levels if captured_noise_levels is not None else noise_levels)[i]
        if compute_log_prob and sched_level > 0:
            if noise_levels[i] > 0:
                lp_c.collect(log_probs[i], i)
            else:
                if dynamics == "CPS":
                    raise NotImplementedError("mi355_flow: log-probs of an evaluation-mode CPS step (the reference reports the storage-rounding "
                                              "residual -mean((round(mu) - mu)^2)) are not produced by the engine")
                like = log_probs[i]
                fill = 0.0 if dynamics == "ODE" else float("nan")
                lp_c.collect(torch.full((like.shape[0],), fill, dtype=torch.float32, device=like.device), i)
        cb_c.collect_step(step_idx=i, output=step_outputs[i] if step_outputs is not None else None, keys=list(extra_keys),
                          capturable={"noise_level": (captured_noise_levels if captured_noise_levels is not None else noise_levels)[i]})
    out = CollectedRollout()
    lats = lat_c.get_result()
    lps = lp_c.get_result() if compute_log_prob else None
    out.latents = torch.stack(lats, dim=1) if lats else None            # (B, P, ...)
    out.log_probs = torch.stack(lps, dim=1) if lps else None            # (B, P')
    out.latent_index_map = lat_c.get_index_map()
    out.log_prob_index_map = lp_c.get_index_map() if compute_log_prob else None
    out.callbacks, out.callback_index_map = cb_c.get_result(), cb_c.get_index_map()
    return out
