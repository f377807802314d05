"""GRPO advantages for group-contiguous data-parallel rollouts (host side, numpy f64).

Mirror of the reference's `AdvantageProcessor.compute_weighted_sum / compute_gdpo` on the
`group_contiguous` path (reference src/flow_factory/advantage/advantage_processor.py:314-481):
group statistics are local (all K repeats of a prompt live on one rank), the only cross-rank
traffic is ONE all-reduce of (n, sum, sum_sq) for the global std (:236-259) -- issued through
`torch.distributed` (RCCL on MI355X, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def _global_mean_std(values: np.ndarray, device: Optional[torch.device] = None):
    t = torch.tensor([float(len(values)), float(np.sum(values)), float(np.sum(values ** 2))], device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    n, s, ss = (float(x) for x in t.tolist())
    mean = s / n
    return mean, max((ss / n - mean ** 2) ** 0.5, 1e-6)


def _group_indices(unique_ids: Sequence[int]) -> np.ndarray:
    return np.unique(np.asarray(unique_ids, dtype=np.int64), return_inverse=True)[1]


def compute_weighted_sum(rewards: Dict[str, np.ndarray], reward_weights: Dict[str, float], unique_ids: Sequence[int],
                         group_size: int, global_std: bool = True, device: Optional[torch.device] = None) -> torch.Tensor:
    gidx = _group_indices(unique_ids)
    agg = np.zeros(len(gidx), dtype=np.float64)
    for k, r in rewards.items():
        agg += np.asarray(r) * reward_weights[k]  # reward dtype (fp32 from torch) times python float, as the reference
    adv = np.zeros_like(agg)
    std = _global_mean_std(agg, device)[1] if global_std else None
    for g in np.unique(gidx):
        m = gidx == g
        gr = agg[m]
        if len(gr) != group_size:
            raise RuntimeError(f"Group size mismatch: expected {group_size}, got {len(gr)} for group {g}")
        s = std if global_std else max(float(np.std(gr)), 1e-6)
        adv[m] = (gr - np.mean(gr)) / s
    return torch.as_tensor(adv)


def compute_gdpo(rewards: Dict[str, np.ndarray], reward_weights: Dict[str, float], unique_ids: Sequence[int],
                 device: Optional[torch.device] = None) -> torch.Tensor:
    gidx = _group_indices(unique_ids)
    parts = []
    for k, r in rewards.items():
        r = np.asarray(r)
        a = np.zeros(len(r), dtype=np.float64)
        for g in np.unique(gidx):
            m = gidx == g
            gr = r[m]
            a[m] = (gr - np.mean(gr)) / max(float(np.std(gr)), 1e-6)
        parts.append(a * reward_weights[k])
    comb = np.sum(parts, axis=0)
    mean, std = _global_mean_std(comb, device)
    return torch.as_tensor((comb - mean) / std)
