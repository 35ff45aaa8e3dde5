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


def broadcast_model_bytes(model_bytes: Optional[bytes], src: int = 0, device=None, force: bool = False) -> bytes:
    """Rank `src` passes the model file bytes, the others None; everyone returns the same bytes.
    Two collectives: the length (int64) and the blob (uint8), both `dist.broadcast` (RCCL over xGMI on GPUs)."""
    import torch
    import torch.distributed as dist
    if not _collectives_wanted(force):
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


def _collectives_wanted(force: bool) -> bool:
    """The collectives run when there is a process group of more than one rank -- or of one rank when `force` (or
    `VPT_DIST_FORCE_COLLECTIVES=1`) says so: a one-GPU box can then push every collective of the path through RCCL once."""
    import os
    import torch.distributed as dist
    if not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or force or os.environ.get("VPT_DIST_FORCE_COLLECTIVES") == "1"


def _agree(ok: bool, dev) -> bool:
    """True when EVERY rank says ok (one all_reduce MIN): a rank never leaves a path on its own -- the next path's collectives
    need everybody (a rank-local failure used to send that rank alone into the staged path's broadcasts)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def broadcast_predictor(predictor, src: int = 0, device=None, model=None, model_bytes: Optional[bytes] = None, force: bool = False,
                        loopback: bool = False):
    """Rank `src` passes its compiled Predictor, the others None; every rank returns a predictor on ITS device holding the
    same tables, and `predictor.tables_broadcast` says how they got there.  Three ways, tried in this order (`VPT_TABLES_BROADCAST=view|
    staged|compile` starts further down; every rank must see the same value):

      view     the table arena is broadcast straight out of rank src's device memory -- a zero-copy torch view of memory the LIBRARY
               allocated (CUDA array interface) -- and adopted with a device-to-device copy (vpt_predictor_describe /
               vpt_predictor_adopt_device): one `dist.broadcast` of a few hundred MB, RCCL over xGMI; nobody but src compiles.
      staged   when that view cannot be made or a rank cannot adopt it: the compiled form goes through memory TORCH owns --
               vpt_predictor_save on src, a torch tensor on the device, `dist.broadcast`, vpt_predictor_load on the others.
      compile  when no big collective works at all: every rank compiles the model itself from `model_bytes` (which the caller has
               broadcast, or read from disk) with src's `predict_tags` (it travels in the header): slower to start, nothing to go wrong.

    `device`: this rank's torch device ("cpu" with the gloo backend and the emulated library of the tests, whose device memory is host
    memory).  The ranks AGREE on leaving a path: src's failed view is announced in the header every rank receives first; after the
    local steps of a path (allocations, adopt, load) an all_reduce(MIN) of every rank's ok flag decides for all of them, src included,
    whether the path stands.  A collective that itself raises on this rank cannot be agreed on (the others are inside it): it is re-raised.
    `force`: run the collectives in a group of ONE rank too (the one-GPU RCCL test).  `loopback`: src also does what a receiver does --
    adopts / loads / compiles from what it broadcast -- and returns THAT predictor, so a group of one exercises the receiving side."""
    import os
    import torch
    import torch.distributed as dist
    from . import _lib, api
    if not _collectives_wanted(force):
        assert predictor is not None
        predictor.tables_broadcast = None
        return predictor
    L = _lib.load()
    dev = device if device is not None else torch.device("cpu")
    rank = dist.get_rank()
    receiver = rank != src or loopback
    dev_index = dev.index if (dev.type == "cuda" and dev.index is not None) else 0
    first = os.environ.get("VPT_TABLES_BROADCAST", "view")
    if first not in TABLES_BROADCAST_PATHS:
        raise ValueError("VPT_TABLES_BROADCAST: one of %s" % (TABLES_BROADCAST_PATHS,))
    start = TABLES_BROADCAST_PATHS.index(first)
    errors = []

    def done(p, path):
        p.tables_broadcast = path if not errors else "%s (after: %s)" % (path, "; ".join(errors))
        return p

    def note(path, e):
        errors.append("%s: %s: %s" % (path, type(e).__name__, str(e)[:120]))

    # ---- what every path needs of src: the sizes, whether the view exists, and src's predict_tags
    meta_len, d_arena, arena_len = C.c_size_t(0), C.c_void_p(), C.c_size_t(0)
    meta, arena, view_ok, tags_flag = np.zeros(1, dtype=np.uint8), None, 0, 0
    if rank == src:
        tags_flag = int(bool(predictor.info()["predict_tags"]))
        if start == 0:
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
                view_ok = 1
            except Exception as e:   # noqa: BLE001 -- whatever it is, the next path does not need the view
                note("view", e)
    header = torch.tensor([meta_len.value, arena_len.value, view_ok, tags_flag], dtype=torch.int64, device=dev)
    dist.broadcast(header, src)
    n_meta, n_arena, view_ok, tags_flag = (int(x) for x in header.tolist())

    # ---- view: the arena as it lies in src's device memory
    if start == 0 and not view_ok and rank != src:
        errors.append("view: refused on rank %d" % src)
    if start == 0 and view_ok:
        t_meta = recv = None
        try:                                                       # local: the receiving buffers
            t_meta = torch.empty(n_meta, dtype=torch.uint8, device=dev)
            if rank == src:
                t_meta.copy_(torch.from_numpy(meta))
            else:
                recv = torch.empty(n_arena, dtype=torch.uint8, device=dev)
            ok = True
        except Exception as e:   # noqa: BLE001
            note("view", e)
            ok = False
        if _agree(ok, dev):
            dist.broadcast(t_meta, src)                            # a raise in here is every rank's (or nobody can help it)
            dist.broadcast(arena if rank == src else recv, src)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            adopted = None
            try:                                                   # local: the adopt
                if receiver:
                    got = recv if rank != src else arena.clone()   # loopback: src adopts from memory torch owns, as a receiver would
                    meta_b = t_meta.cpu().numpy()
                    h = C.c_void_p()
                    st = L.vpt_predictor_adopt_device(meta_b.ctypes.data, meta_b.nbytes, got.data_ptr(), n_arena, dev_index, C.byref(h))
                    if st != _lib.VPT_OK:
                        api._raise(st)
                    adopted = api.Predictor._adopt(h, model, dev_index)
                ok = True
            except Exception as e:   # noqa: BLE001
                note("view", e)
                ok = False
            if _agree(ok, dev):
                return done(adopted if receiver else predictor, "view")
            if ok:
                errors.append("view: another rank could not adopt")
        elif ok:
            errors.append("view: another rank could not allocate")

    # ---- staged: the compiled form through torch-owned memory
    if start <= 1:
        blob = t = None
        try:                                                       # local: src serialises
            if rank == src:
                blob = np.frombuffer(predictor.save_compiled(), dtype=np.uint8)
            ok = True
        except Exception as e:   # noqa: BLE001
            note("staged", e)
            ok = False
        n = torch.tensor([len(blob) if (rank == src and ok) else -1], dtype=torch.int64, device=dev)
        dist.broadcast(n, src)                                     # -1: src could not serialise -- everybody moves on
        if int(n.item()) >= 0:
            try:                                                   # local: the buffer
                t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
                if rank == src:
                    t.copy_(torch.from_numpy(blob.copy()))
                ok = True
            except Exception as e:   # noqa: BLE001
                note("staged", e)
                ok = False
            if _agree(ok, dev):
                dist.broadcast(t, src)
                loaded = None
                try:                                               # local: the load
                    if receiver:
                        loaded = api.Predictor.load_compiled(t.cpu().numpy().tobytes(), model=model, device=dev_index)
                    ok = True
                except Exception as e:   # noqa: BLE001
                    note("staged", e)
                    ok = False
                if _agree(ok, dev):
                    return done(loaded if receiver else predictor, "staged")
                if ok:
                    errors.append("staged: another rank could not load")
            elif ok:
                errors.append("staged: another rank could not allocate")
        elif rank != src:
            errors.append("staged: refused on rank %d" % src)

    # ---- compile: every rank for itself, with src's predict_tags
    if not receiver:
        return done(predictor, "compile")
    if model_bytes is None and model is None:
        raise RuntimeError("broadcast_predictor: no collective path worked (%s) and no model_bytes were given to compile from" % "; ".join(errors))
    m = model if model is not None else api.Model.read_slice(model_bytes)[0]
    return done(api.Predictor(m, bool(tags_flag), device=dev_index), "compile")


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


def reduce_throughput(elapsed_s: float, units: float, device=None, force: bool = False) -> Tuple[float, float]:
    """(max elapsed over ranks, total units over ranks) -- the bench contract's whole-job figures."""
    import torch
    import torch.distributed as dist
    if not _collectives_wanted(force):
        return elapsed_s, units
    dev = device if device is not None else torch.device("cpu")
    el = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    tot = torch.tensor([units], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(el.item()), float(tot.item())
