"""Multi-GPU: one process per GPU, population tree-sharded, one RCCL collective.

The path shards embarrassingly: trees are independent given X.  Rank r owns
trees {t : t mod world == r} (round-robin balances node counts), X is replicated, every rank
writes its own [n_trees/world, N] output slab locally, and the only exchange is an
all_gather of the per-tree completion flags (n_trees bytes; latency-bound over xGMI).  Full
outputs are never gathered: at config C4 they are 400 GB, more than one GPU's HBM.

backend "nccl" == RCCL on ROCm; the CPU tests run the same code over gloo.
"""
from __future__ import annotations

from typing import List

import numpy as np


def shard_indices(n_trees: int, rank: int, world: int) -> List[int]:
    """Global tree ids owned by `rank` (round-robin)."""
    return list(range(rank, n_trees, world))


def shard_size(n_trees: int, rank: int, world: int) -> int:
    return (n_trees - rank + world - 1) // world if n_trees > rank else 0


def gather_flags(local_ok, n_trees: int, rank: int, world: int):
    """all_gather of the per-tree `ok` bytes; returns the flags of ALL trees in global tree order
    (uint8 tensor of length n_trees on the caller's device)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local_ok
    per = (n_trees + world - 1) // world  # padded shard length so all_gather shapes match
    # gloo (CPU tests, and the 1-GPU dry run of bench.py's multi-rank path) gathers host tensors
    dev = local_ok.device if dist.get_backend() != "gloo" else torch.device("cpu")
    buf = torch.ones(per, dtype=torch.uint8, device=dev)
    buf[: local_ok.numel()] = local_ok.to(torch.uint8).to(dev)
    gathered = torch.empty(world * per, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, buf)
    # rank r holds trees r, r+world, ...: entry [r, i] is tree r + i*world
    return gathered.view(world, per).t().reshape(-1)[:n_trees].to(local_ok.device)


def scatter_population(trees, rank: int, world: int):
    return [trees[i] for i in shard_indices(len(trees), rank, world)]


def unshard_order(n_trees: int, world: int) -> np.ndarray:
    """Permutation p with global_tree = p[k] for the concatenation of rank shards."""
    return np.concatenate([np.arange(r, n_trees, world) for r in range(world)]) if n_trees else np.zeros(0, int)
