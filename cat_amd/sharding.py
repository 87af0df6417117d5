"""Batch sharding of independent ciphertext operations over the GPUs of one node.

The path shards without any data-path collective: ciphertext operations are independent units, so rank r
of W takes a contiguous slice of the batch, holds its own replica of the read-only state (NTT tables inside
its Context, relinearization / Galois keys) and never talks to the other ranks while computing.  RCCL over
xGMI (torch.distributed backend "nccl" on ROCm) is used for exactly two things, both outside the steady
state: broadcasting key material from the rank that loaded it, and (optionally) gathering results.

The reference has no counterpart (single GPU, single process: no nccl/MPI anywhere under native/src,
SURVEY.md 2a); this module is the MI355X-side addition that BASELINE config 5 asks for (8192 independent
CKKS N=2^15 mul+relin, 1024 per GPU at 8 GPUs).  It is backend-agnostic so the same code is exercised on
CPU with gloo in tests/test_sharding_gloo.py.
"""
from __future__ import annotations

import ctypes
import glob
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of rank `rank`; the first n_items % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("invalid shard request")
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_range(n_items, world, r)[1] - shard_range(n_items, world, r)[0] for r in range(world)]


def world_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def broadcast_key(key: torch.Tensor, src: int = 0, bucket_bytes: int = 256 << 20) -> torch.Tensor:
    """Replicates a key tensor ([k-1, 2, k, N] words) from `src` to every rank, in place.

    One-to-all over xGMI is per-link bound (7 links x ~153 GB/s per GPU), so the key is sent in a few large
    buckets rather than per limb: a 126 MB relinearization key (N=2^15, 16 primes) is one bucket, a 629 MB
    Galois key (N=2^16, 25 primes) three."""
    world, _ = world_info()
    if world == 1:
        return key
    flat = key.view(-1)
    step = max(1, bucket_bytes // flat.element_size())
    via_host = _host_staged(flat)
    for off in range(0, flat.numel(), step):
        piece = flat[off : off + step]
        if via_host:
            staged = piece.cpu()
            dist.broadcast(staged, src=src)
            piece.copy_(staged)
        else:
            dist.broadcast(piece, src=src)
    return key


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]  # ncclUniqueId, rccl.h


class RcclComm:
    """An RCCL communicator of the library's own (one rank per GPU), for the C ABI's one collective, cat_key_broadcast.

    librccl is bound with ctypes -- the copy PyTorch ships when there is one (the process then holds ONE RCCL, loaded with
    RTLD_GLOBAL so that libcat_amd.so's dlsym finds the same instance), else /opt/rocm's.  Rank 0 draws the ncclUniqueId
    and shares it through the process group that already exists (any backend); ncclCommInitRank then connects the
    ranks over xGMI.  `handle` is the ncclComm_t to pass to Context.key_broadcast.  A one-GPU box can only make a
    one-rank communicator (RCCL refuses two ranks on one device); the N > 1 path first runs on a multi-GPU node."""

    def __init__(self, world: int, rank: int):
        self.lib = self._load()
        uid = _NcclUniqueId()
        if rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            # the raw 128 bytes of the id (a c_char array read as .value would stop at the first NUL)
            box = [ctypes.string_at(ctypes.addressof(uid), 128) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctypes.memmove(ctypes.addressof(uid), box[0], 128)
        self.handle = ctypes.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
        self._check(self.lib.ncclCommInitRank(ctypes.byref(self.handle), world, uid, rank), "ncclCommInitRank")
        self.world, self.rank = world, rank

    @staticmethod
    def _load():
        names = sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")))
        names += ["librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"]
        last = None
        for name in names:
            try:
                return ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
            except OSError as e:
                last = e
        raise RuntimeError(f"librccl.so is not available: {last}")

    def _check(self, rc: int, what: str):
        if rc != 0:
            self.lib.ncclGetErrorString.restype = ctypes.c_char_p
            raise RuntimeError(f"RCCL error in {what}: {self.lib.ncclGetErrorString(rc).decode()}")

    def close(self):
        if self.handle:
            self.lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            self.lib.ncclCommDestroy(self.handle)
            self.handle = ctypes.c_void_p()


def _host_staged(t: torch.Tensor) -> bool:
    """gloo moves device tensors through host memory anyway; doing the copies here keeps its stream handling out of
    the picture (rehearsals of the multi-rank path on one GPU, tests)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def gather_results(local: torch.Tensor, n_items: int, dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """Concatenates the per-rank result slices (dim 0) in rank order.

    dst=None: every rank gets the full batch (all_gather); otherwise only rank `dst` (others return None).
    Slices may differ by one item (shard_range), so shorter slices are padded for the collective."""
    world, rank = world_info()
    if world == 1:
        return local
    sizes = shard_sizes(n_items, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError("local slice does not match shard_range")
    width = max(sizes)
    if local.shape[0] < width:
        pad = torch.zeros((width - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    if dst is None:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
    else:
        parts = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
        dist.gather(local, parts, dst=dst)
        if rank != dst:
            return None
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def run_sharded(op: Callable[[int, int], torch.Tensor], n_items: int, gather: bool = True,
                dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """Runs `op(begin, end)` on this rank's slice of an n_items batch and optionally gathers the results.

    `op` is the single-GPU pipeline (e.g. Context.ckks_multiply + relinearize on items [begin, end)); there is
    no communication inside it."""
    world, rank = world_info()
    begin, end = shard_range(n_items, world, rank)
    local = op(begin, end)
    if not gather:
        return local
    return gather_results(local, n_items, dst=dst)


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """Timing helper for the bench contract: the slowest rank's time."""
    world, _ = world_info()
    if world == 1:
        return value
    if dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
