#!/usr/bin/env python3
"""Forward / inverse NTT times by ring size (N = 4096 ... 32768), plain batched transforms of ~1 GB.
Usage: ntt_sizes.py [log_n ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import cat_amd
from bench import get_primes_desc

for log_n in ([int(a) for a in sys.argv[1:]] or (12, 13, 14, 15)):
    n = 1 << log_n
    limbs = 16
    primes = get_primes_desc(2 * n, 55, limbs)
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_NONE, device=0)
    polys = (1 << 30) // (limbs * n * 8)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(1)
    data = torch.empty((polys, limbs, n), dtype=torch.int64, device="cuda:0")
    for j, q in enumerate(primes):
        data[:, j, :] = torch.randint(0, q, (polys, n), dtype=torch.int64, device="cuda:0", generator=gen)
    ref = data.clone()
    best_f = best_i = 1e9
    for _ in range(5):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ctx.ntt_forward(data)
        e1.record()
        ctx.ntt_inverse(data)
        e2.record()
        torch.cuda.synchronize()
        best_f, best_i = min(best_f, e0.elapsed_time(e1)), min(best_i, e1.elapsed_time(e2))
    assert torch.equal(data, ref)
    alg = 16 * data.numel()
    print(f"N={n:6d} {limbs} x 55-bit primes, {polys} polys ({data.numel() * 8 / 2**30:.2f} GiB): forward {best_f:.3f} ms = {alg / best_f / 1e6:.0f} GB/s, "
          f"inverse {best_i:.3f} ms = {alg / best_i / 1e6:.0f} GB/s (algorithmic, 16 B per coefficient)")
    ctx.close()
