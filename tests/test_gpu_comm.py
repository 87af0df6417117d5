"""cat_key_broadcast (include/cat_amd.h): key replication over RCCL, the path's only collective.
A one-GPU box can hold only a one-rank communicator (RCCL refuses two ranks on one device), so this covers the binding
of librccl at run time, the bucketing and the error path; the multi-rank data movement is the same ncclBroadcast that
tests/test_sharding_gloo.py exercises through torch.distributed with world size 2."""
import ctypes as C

import pytest
import torch

import cat_amd

pytestmark = pytest.mark.gpu


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_key_broadcast_single_rank_communicator_and_no_comm():
    n, log_n = 4096, 12
    import bench

    primes = list(reversed(bench.get_primes_desc(2 * n, 50, 3)))
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_CKKS, device=0)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(1)
    key = torch.randint(0, min(primes), (2, 2, 3, n), dtype=torch.int64, device="cuda:0", generator=gen)
    ref = key.clone()
    ctx.key_broadcast(key)  # no communicator: a single process already holds the key
    assert torch.equal(key, ref)
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1")
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        # several buckets, the last one ragged
        ctx.key_broadcast(key, comm, root=0, bucket_words=5000)
        torch.cuda.synchronize()
        assert torch.equal(key, ref)
        with pytest.raises(RuntimeError, match="RCCL error"):
            ctx.key_broadcast(key, comm, root=3)  # no such rank
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
    ctx.close()


def test_bench_communicator_class_on_one_rank():
    """cat_amd.sharding.RcclComm (what bench.py --gpus N hands to cat_key_broadcast): world size 1 needs no process group;
    the handle it makes is a valid ncclComm_t for the C ABI's collective, which binds the SAME RCCL instance (dlsym of the
    copy loaded RTLD_GLOBAL here before any of its own).  The N > 1 exchange needs a multi-GPU node."""
    import bench
    from cat_amd import sharding

    n, log_n = 4096, 12
    primes = list(reversed(bench.get_primes_desc(2 * n, 50, 3)))
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_CKKS, device=0)
    comm = sharding.RcclComm(1, 0)
    try:
        gen = torch.Generator(device="cuda:0")
        gen.manual_seed(2)
        key = torch.randint(0, min(primes), (2, 2, 3, n), dtype=torch.int64, device="cuda:0", generator=gen)
        ref = key.clone()
        ctx.key_broadcast(key, comm.handle, root=0)
        torch.cuda.synchronize()
        assert torch.equal(key, ref)
        # the second collective: one rank's slice is the whole batch (ncclAllGather on the one-rank communicator, and the
        # single-process form without one)
        part = key[:1].contiguous()
        got = ctx.result_gather(part, comm.handle, world=1)
        alone = ctx.result_gather(part, None, world=1)
        torch.cuda.synchronize()
        assert got.shape == (1,) + tuple(part.shape) and torch.equal(got[0], part) and torch.equal(alone[0], part)
        with pytest.raises(ValueError, match="world x local"):
            ctx.result_gather(part, None, world=1, out=torch.empty(3, dtype=torch.int64, device="cuda:0"))
    finally:
        comm.close()
    ctx.close()


def test_bench_guarded_product_broadcast_runs_on_a_one_rank_communicator():
    """bench.py's N > 1 key broadcast runs in a worker thread with a time limit and falls back to torch.distributed when
    it does not complete on every rank.  --rccl-selftest drives that code with one rank (its own device binding, the
    communicator, cat_key_broadcast from a second thread); the line must say the product's collective carried the key."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--rccl-selftest", "--steps", "2", "--warmup", "1", "--settle", "0",
                        "--skip-ntt", "--skip-cpu", "--skip-latency", "--batch", "2", "--limbs", "4"],
                       cwd=root, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["key_broadcast_via"].startswith("cat_key_broadcast"), line["config"]["key_broadcast_via"]
    assert line["parity_checked"] is True
