"""Multi-GPU layout of the likelihood path: chains shard, nothing else does.

Chains (and the proposals of distinct chains) are fully independent, so one process per GPU owns a
contiguous range of chains -- its thetas, RNG streams, adaptation state and log-posteriors -- and
the sampling loop contains NO collective.  The only exchange is the optional post-calibration
all-gather of fixed-width per-chain summary records, which lets every rank
compute the ensemble quantiles the reference computes serially in
src/model/ResultAggregator.cpp:35-172.  ``torch.distributed`` backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(n_chains: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of the chains owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_chains, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chain_seed(seed0: int, global_chain: int) -> int:
    """std::mt19937 seed of a chain: independent of how chains are sharded."""
    return (seed0 + global_chain) & 0xFFFFFFFF


def summary_record(samples: np.ndarray, values: np.ndarray, accepted: np.ndarray) -> np.ndarray:
    """[P posterior means | P variances | best log-posterior | accept count] per chain.

    samples: C x S x P, values: C x S, accepted: C  ->  C x (2P + 2) float64."""
    mean = samples.mean(axis=1)
    var = samples.var(axis=1, ddof=1) if samples.shape[1] > 1 else np.zeros_like(mean)
    return np.concatenate([mean, var, values.max(axis=1, keepdims=True),
                           accepted.astype(np.float64)[:, None]], axis=1)


def all_gather_records(local: "np.ndarray | object", n_total: int, device=None):
    """All-gather per-chain records of unequal shard sizes (pads to the largest shard).

    Works on any initialised torch.distributed backend; returns an (n_total x W) torch tensor on
    every rank, ordered by global chain index."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    width = t.shape[1]
    max_rows = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    padded = torch.zeros(max_rows, width, dtype=t.dtype, device=t.device)
    padded[:t.shape[0]] = t
    out = torch.empty(world * max_rows, width, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, padded)
    pieces = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        pieces.append(out[r * max_rows:r * max_rows + (hi - lo)])
    return torch.cat(pieces, dim=0)


def ensemble_quantiles(records, probs=(0.025, 0.5, 0.975)):
    """Exact sort-based quantiles across chains of every record column (the reference's choice for
    trajectories, PostCalibrationAnalyser.cpp:303-340; P-square would be order dependent)."""
    import torch
    q = torch.tensor(probs, dtype=records.dtype, device=records.device)
    return torch.quantile(records, q, dim=0)
