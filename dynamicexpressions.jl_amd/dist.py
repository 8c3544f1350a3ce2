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


class Comm:
    """The same exchange through the C ABI (``de_dist_*``, csrc/de_dist.cpp: RCCL loaded by the library itself) — what a
    Julia or C caller uses; no torch.distributed involved.  ``unique_id()`` on rank 0, ship the 128 bytes to the other
    ranks by any means, ``Comm(ctx, rank, world, id)`` everywhere."""

    def __init__(self, ctx, rank: int = 0, world: int = 1, unique_id: bytes = b""):
        import ctypes as C
        from . import api
        self._lib, self._ctx, self.rank, self.world = api.library(), ctx, rank, world
        self._h = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(unique_id), 128) if world > 1 else None
        rc = self._lib.de_dist_init(ctx._h, rank, world, idbuf, C.byref(self._h))
        if rc != 0:
            raise api.DeviceError(f"de_dist_init: {self._lib.de_status_string(rc).decode()}: {self._lib.de_dist_last_error(None).decode()}")

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import api
        lib = api.library()
        buf = C.create_string_buffer(128)
        rc = lib.de_dist_unique_id(buf)
        if rc != 0:
            raise api.DeviceError(f"de_dist_unique_id: {lib.de_dist_last_error(None).decode()}")
        return buf.raw

    def _check(self, rc: int) -> None:
        if rc != 0:
            from . import api
            raise api.DeviceError(f"{self._lib.de_status_string(rc).decode()}: {self._lib.de_dist_last_error(self._h).decode()}")

    def set_timeout(self, timeout_ms: int) -> None:
        """Bound every collective of this communicator: the calls then WAIT for what they queued and raise DeviceError (DE_ERR_RCCL, the
        communicator aborted) after ``timeout_ms`` instead of hanging on a peer that is down; 0 = asynchronous calls (the default)."""
        self._check(self._lib.de_dist_set_timeout(self._h, int(timeout_ms)))

    def world_size(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount)."""
        return int(self._lib.de_dist_world_size(self._h))

    def shard_size(self, n_trees: int) -> int:
        return int(self._lib.de_dist_shard_size(n_trees, self.rank, self.world))

    def broadcast(self, tensor, root: int = 0) -> None:
        """Replicate a device tensor (X) from ``root``; asynchronous on the context's stream."""
        self._check(self._lib.de_dist_broadcast(self._h, tensor.data_ptr(), tensor.numel() * tensor.element_size(), root))

    def gather_flags(self, ok_local, n_trees: int):
        """Flags of ALL trees in global tree order (uint8 device tensor of length n_trees) from every rank's local flags."""
        import torch
        out = torch.empty(n_trees, dtype=torch.uint8, device=ok_local.device)
        loc = ok_local.to(torch.uint8).contiguous()
        self._check(self._lib.de_dist_gather_flags(self._h, loc.data_ptr(), n_trees, out.data_ptr()))
        return out

    def close(self) -> None:
        if self._h:
            self._lib.de_dist_destroy(self._h)
            self._h = None
