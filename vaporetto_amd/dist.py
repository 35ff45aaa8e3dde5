"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm, "gloo" in the
CPU tests).  Sentences are independent (predictor.rs:518-543 carries no state between calls), so the only
collective on the path is the one-time broadcast of the model file from rank 0; every rank then scores its own
contiguous shard of the batch and there is no data-path exchange.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def broadcast_model_bytes(model_bytes: Optional[bytes], src: int = 0, device=None) -> bytes:
    """Rank `src` passes the model file bytes, the others None; everyone returns the same bytes.
    Two collectives: the length (int64) and the blob (uint8), both `dist.broadcast` (RCCL over xGMI on GPUs)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert model_bytes is not None
        return model_bytes
    dev = device if device is not None else torch.device("cpu")
    rank = dist.get_rank()
    n = torch.tensor([len(model_bytes) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    blob = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == src:
        blob.copy_(torch.frombuffer(bytearray(model_bytes), dtype=torch.uint8))
    dist.broadcast(blob, src)
    return blob.cpu().numpy().tobytes()


def shard_bounds(byte_offsets: np.ndarray, world: int) -> np.ndarray:
    """Contiguous sentence ranges balanced by BYTE count (a proxy for chars that needs no decode; it matters
    for ragged batches such as configs[4]).  Returns int64[world + 1]: rank r owns sentences [b[r], b[r+1])."""
    boff = np.asarray(byte_offsets, dtype=np.uint64)
    S = len(boff) - 1
    total = int(boff[S] - boff[0])
    targets = int(boff[0]) + (np.arange(1, world, dtype=np.float64) * total / world)
    cuts = np.searchsorted(boff.astype(np.float64), targets, side="left").astype(np.int64)
    bounds = np.concatenate([[0], np.clip(cuts, 0, S), [S]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def take_shard(utf8: np.ndarray, byte_offsets: np.ndarray, rank: int, world: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """This rank's slice of a packed batch: (utf8 slice, rebased byte offsets, index of its first sentence)."""
    b = shard_bounds(byte_offsets, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    boff = np.asarray(byte_offsets, dtype=np.uint64)
    t0, t1 = int(boff[lo]), int(boff[hi])
    return np.ascontiguousarray(utf8[t0:t1]), (boff[lo:hi + 1] - boff[lo]).astype(np.uint64), lo


def reduce_throughput(elapsed_s: float, units: float, device=None) -> Tuple[float, float]:
    """(max elapsed over ranks, total units over ranks) -- the bench contract's whole-job figures."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed_s, units
    dev = device if device is not None else torch.device("cpu")
    el = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    tot = torch.tensor([units], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(el.item()), float(tot.item())
