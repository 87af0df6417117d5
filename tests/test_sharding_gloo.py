"""The N>1 host path on CPU: world_size-2 gloo run of the batch sharding used by bench.py --gpus N and by
BASELINE config 5 (independent ciphertext pairs, contiguous slices per rank, key broadcast once, result
gather).  The per-rank compute step is played by the CPU oracle here (tests may use it; the product's
compute step is the HIP library and is covered by the -m gpu tests): what this file checks is that
partition + broadcast + gather reproduce the single-process result bit for bit, including uneven slices."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cat_amd import sharding


def test_shard_range_partitions_exactly():
    for n in (0, 1, 5, 8, 1024, 8192, 8193):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(n, w)
    assert sharding.shard_sizes(8192, 8) == [1024] * 8  # BASELINE config 5
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import seal_oracle as so

        log_n, n = 10, 1024
        primes = so.coeff_modulus_create(n, [40, 30, 30, 40])
        octx = so.Context(log_n, primes, so.SCHEME_CKKS)
        L = 3
        # rank 0 "loads" the key; everyone else receives it over the collective
        key = torch.zeros((len(primes) - 1, 2, len(primes), n), dtype=torch.int64)
        if rank == 0:
            sk = octx.secret_key(1)
            key.copy_(torch.from_numpy(octx.relin_key(sk, 2).view(np.int64)))
        sharding.broadcast_key(key, src=0, bucket_bytes=64 << 10)  # several buckets
        hkey = key.numpy().view(np.uint64)

        def item(i):
            x, y = octx.uniform(L, 2, 100 + i), octx.uniform(L, 2, 200 + i)
            return octx.rescale(octx.relinearize(octx.ckks_multiply(x, y), hkey))

        def op(begin, end):
            if end == begin:
                return torch.empty((0, 2, L - 1, n), dtype=torch.int64)
            return torch.from_numpy(np.stack([item(i) for i in range(begin, end)]).view(np.int64))

        full = sharding.run_sharded(op, n_items)                  # all_gather
        on0 = sharding.run_sharded(op, n_items, dst=0)            # gather to rank 0
        t = sharding.max_over_ranks(float(rank + 1))
        assert t == float(world)
        assert (on0 is None) == (rank != 0)
        np.save(os.path.join(out_dir, f"full{rank}.npy"), full.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "on0.npy"), on0.numpy())
            np.save(os.path.join(out_dir, "key.npy"), key.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 2, 1])
def test_two_rank_gloo_matches_single_process(tmp_path, oracle, n_items):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    so = oracle
    log_n, n = 10, 1024
    primes = so.coeff_modulus_create(n, [40, 30, 30, 40])
    octx = so.Context(log_n, primes, so.SCHEME_CKKS)
    hkey = octx.relin_key(octx.secret_key(1), 2)
    assert (np.load(tmp_path / "key.npy").view(np.uint64) == hkey).all()
    exp = np.stack([
        octx.rescale(octx.relinearize(octx.ckks_multiply(octx.uniform(3, 2, 100 + i), octx.uniform(3, 2, 200 + i)), hkey))
        for i in range(n_items)
    ])
    for name in ("full0.npy", "full1.npy", "on0.npy"):
        got = np.load(tmp_path / name).view(np.uint64)
        assert got.shape == exp.shape and (got == exp).all(), name
