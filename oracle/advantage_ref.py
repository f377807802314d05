"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Numpy-f64 restatement of the reference's GRPO advantage math (parity row P1 of SURVEY.md 8(a)):

  * compute_weighted_sum -- src/flow_factory/advantage/advantage_processor.py:314-397
  * compute_gdpo         -- same file :403-481
  * _global_mean_std     -- same file :236-259  (population std via (n, sum, sum_sq))

Single-process view over the GLOBAL reward arrays: what every rank must agree on after the
reference's all-reduce of (n, sum, sum_sq).  Pinned by tests/golden/advantage_*.npz, generated
by running the reference's own AdvantageProcessor (oracle/make_golden.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def global_mean_std(values: np.ndarray):
    # the reference ships (n, sum, sum_sq) through a default-dtype (float32) torch tensor
    # before the all-reduce (:251-256), so the statistics carry fp32 rounding
    n = float(np.float32(len(values)))
    s = float(np.float32(np.sum(values)))
    ss = float(np.float32(np.sum(values**2)))
    mean = s / n
    std = max((ss / n - mean**2) ** 0.5, 1e-6)
    return mean, std


def group_indices_from_ids(unique_ids) -> np.ndarray:
    _, inv = np.unique(np.asarray(unique_ids, dtype=np.int64), return_inverse=True)
    return inv


def weighted_sum(rewards: Dict[str, np.ndarray], weights: Dict[str, float], group_indices: np.ndarray,
                 group_size: int, global_std: bool = True) -> np.ndarray:
    agg = np.zeros_like(next(iter(rewards.values())), dtype=np.float64)
    for k, r in rewards.items():
        agg += np.asarray(r) * weights[k]
    adv = np.zeros_like(agg, dtype=np.float64)
    if global_std:
        _, std = global_mean_std(agg)
    for g in np.unique(group_indices):
        mask = group_indices == g
        gr = agg[mask]
        if len(gr) != group_size:
            raise RuntimeError(f"Group size mismatch: expected {group_size}, got {len(gr)} for group {g}")
        mean = np.mean(gr, axis=0, keepdims=True)
        if not global_std:
            std = max(np.std(gr, axis=0, keepdims=True), 1e-6)
        adv[mask] = (gr - mean) / std
    return adv


def gdpo(rewards: Dict[str, np.ndarray], weights: Dict[str, float], group_indices: np.ndarray) -> np.ndarray:
    parts = []
    for k, r in rewards.items():
        r = np.asarray(r)
        a = np.zeros_like(r, dtype=np.float64)
        for g in np.unique(group_indices):
            mask = group_indices == g
            gr = r[mask]
            a[mask] = (gr - np.mean(gr)) / max(np.std(gr), 1e-6)
        parts.append(a * weights[k])
    comb = np.sum(parts, axis=0)
    mean, std = global_mean_std(comb)
    return (comb - mean) / std
