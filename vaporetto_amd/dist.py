"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm, "gloo" in the
CPU tests).  Sentences are independent (predictor.rs:518-543 carries no state between calls), so the only
collective on the path is the one-time broadcast of the predictor from rank 0 -- the model file bytes, or better the
COMPILED tables, device to device over xGMI, so that only one rank compiles -- and every rank then scores its own
contiguous shard of the batch, balanced by character count; there is no data-path exchange.
"""
from __future__ import annotations

import ctypes as C
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


class _DeviceBytes:
    """A device address range as an object torch can view without copying (CUDA array interface, which ROCm builds honour)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


TABLES_BROADCAST_PATHS = ("view", "staged", "compile")


def broadcast_predictor(predictor, src: int = 0, device=None, model=None, model_bytes: Optional[bytes] = None, predict_tags: bool = False):
    """Rank `src` passes its compiled Predictor, the others None; every rank returns a predictor on ITS device holding the
    same tables, and `predictor.tables_broadcast` says how they got there.  Three ways, tried in this order (`VPT_TABLES_BROADCAST=view|
    staged|compile` starts further down; every rank must see the same value):

      view     the table arena is broadcast straight out of rank src's device memory -- a zero-copy torch view of memory the LIBRARY
               allocated (CUDA array interface) -- and adopted with a device-to-device copy (vpt_predictor_describe /
               vpt_predictor_adopt_device): one `dist.broadcast` of a few hundred MB, RCCL over xGMI; nobody but src compiles.
      staged   when that view cannot be made or the collective refuses it (the library links /opt/rocm's HIP runtime, torch brings its own:
               the first 8-GPU run is the first time the two meet): the compiled form goes through memory TORCH owns --
               vpt_predictor_save on src, a torch tensor on the device, `dist.broadcast`, vpt_predictor_load on the others.
      compile  when no big collective works at all: every rank compiles the model itself from `model_bytes` (which the caller has
               broadcast, or read from disk): slower to start, nothing to go wrong.

    `device`: this rank's torch device ("cpu" with the gloo backend and the emulated library of the tests, whose device memory is host
    memory).  Which path a failure leads to is decided the same way on every rank: a failed view on src is announced in the header every
    rank receives first; a collective that raises does so on every rank."""
    import os
    import torch
    import torch.distributed as dist
    from . import _lib, api
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert predictor is not None
        predictor.tables_broadcast = None
        return predictor
    L = _lib.load()
    dev = device if device is not None else torch.device("cpu")
    rank = dist.get_rank()
    dev_index = dev.index if (dev.type == "cuda" and dev.index is not None) else 0
    first = os.environ.get("VPT_TABLES_BROADCAST", "view")
    if first not in TABLES_BROADCAST_PATHS:
        raise ValueError("VPT_TABLES_BROADCAST: one of %s" % (TABLES_BROADCAST_PATHS,))
    start = TABLES_BROADCAST_PATHS.index(first)
    errors = []

    def done(p, path):
        p.tables_broadcast = path if not errors else "%s (after: %s)" % (path, "; ".join(errors))
        return p

    # ---- view: the arena as it lies in src's device memory
    if start == 0:
        meta_len, d_arena, arena_len = C.c_size_t(0), C.c_void_p(), C.c_size_t(0)
        meta, arena, view_ok = np.zeros(1, dtype=np.uint8), None, 1
        if rank == src:
            try:
                st = L.vpt_predictor_describe(predictor.handle, None, 0, C.byref(meta_len), C.byref(d_arena), C.byref(arena_len))
                if st != _lib.VPT_OK:
                    api._raise(st)
                meta = np.zeros(meta_len.value, dtype=np.uint8)
                st = L.vpt_predictor_describe(predictor.handle, meta.ctypes.data, meta.nbytes, C.byref(meta_len), C.byref(d_arena), C.byref(arena_len))
                if st != _lib.VPT_OK:
                    api._raise(st)
                if dev.type == "cuda":
                    arena = torch.as_tensor(_DeviceBytes(d_arena.value, arena_len.value), device=dev)     # a view of the predictor's own tables
                    int(arena[-1].item())                                                                 # ... that torch can really read
                else:
                    arena = torch.from_numpy(np.ctypeslib.as_array(C.cast(d_arena, C.POINTER(C.c_uint8)), shape=(arena_len.value,)))
            except Exception as e:   # noqa: BLE001 -- whatever it is, the next path does not need the view
                view_ok = 0
                errors.append("view: %s: %s" % (type(e).__name__, str(e)[:120]))
        header = torch.tensor([meta_len.value, arena_len.value, view_ok], dtype=torch.int64, device=dev)
        dist.broadcast(header, src)
        n_meta, n_arena, view_ok = (int(x) for x in header.tolist())
        if view_ok:
            try:
                t_meta = torch.empty(n_meta, dtype=torch.uint8, device=dev)
                if rank == src:
                    t_meta.copy_(torch.from_numpy(meta))
                dist.broadcast(t_meta, src)
                if rank != src:
                    arena = torch.empty(n_arena, dtype=torch.uint8, device=dev)
                dist.broadcast(arena, src)
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                if rank == src:
                    return done(predictor, "view")
                meta_b = t_meta.cpu().numpy()
                h = C.c_void_p()
                st = L.vpt_predictor_adopt_device(meta_b.ctypes.data, meta_b.nbytes, arena.data_ptr(), n_arena, dev_index, C.byref(h))
                if st != _lib.VPT_OK:
                    api._raise(st)
                return done(api.Predictor._adopt(h, model, dev_index), "view")
            except Exception as e:   # noqa: BLE001
                errors.append("view: %s: %s" % (type(e).__name__, str(e)[:120]))
        elif rank != src:
            errors.append("view: refused on rank %d" % src)

    # ---- staged: the compiled form through torch-owned memory
    if start <= 1:
        try:
            blob = None
            if rank == src:
                blob = np.frombuffer(predictor.save_compiled(), dtype=np.uint8)
            n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
            dist.broadcast(n, src)
            t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
            if rank == src:
                t.copy_(torch.from_numpy(blob.copy()))
            dist.broadcast(t, src)
            if rank == src:
                return done(predictor, "staged")
            return done(api.Predictor.load_compiled(t.cpu().numpy().tobytes(), model=model, device=dev_index), "staged")
        except Exception as e:   # noqa: BLE001
            errors.append("staged: %s: %s" % (type(e).__name__, str(e)[:120]))

    # ---- compile: every rank for itself
    if rank == src:
        return done(predictor, "compile")
    if model_bytes is None:
        raise RuntimeError("broadcast_predictor: no collective path worked (%s) and no model_bytes were given to compile from" % "; ".join(errors))
    m = model if model is not None else api.Model.read_slice(model_bytes)[0]
    return done(api.Predictor(m, predict_tags, device=dev_index), "compile")


def shard_bounds(out_offsets: np.ndarray, world: int) -> np.ndarray:
    """Contiguous sentence ranges balanced by CHARACTER count (what scoring costs; bytes per char vary 1..4 in mixed
    text, so byte balance can be off by 3x).  `out_offsets` = vpt_count_boundaries' output (boundaries before sentence i);
    chars before sentence i = out_offsets[i] + i.  Returns int64[world + 1]: rank r owns sentences [b[r], b[r+1])."""
    from . import api
    return api.shard_bounds(out_offsets, world).astype(np.int64)


def take_shard(utf8: np.ndarray, byte_offsets: np.ndarray, out_offsets: np.ndarray, rank: int, world: int):
    """This rank's slice of a packed batch: (utf8 slice, rebased byte offsets, rebased out offsets, index of its first sentence)."""
    b = shard_bounds(out_offsets, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    boff = np.asarray(byte_offsets, dtype=np.uint64)
    ooff = np.asarray(out_offsets, dtype=np.uint64)
    t0, t1 = int(boff[lo]), int(boff[hi])
    return (np.ascontiguousarray(utf8[t0:t1]), (boff[lo:hi + 1] - boff[lo]).astype(np.uint64),
            (ooff[lo:hi + 1] - ooff[lo]).astype(np.uint64), lo)


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
