#!/usr/bin/env python3
"""Does the base address of the buffer matter?  Forward / inverse NTT of the same 1 GiB batch placed at different byte
offsets inside one allocation (and in a second, later allocation).  Usage: ntt_offset.py [log_n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import cat_amd
from bench import get_primes_desc

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n = 1 << log_n
limbs = 16
primes = get_primes_desc(2 * n, 55, limbs)
ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_NONE, device=0)
polys = (1 << 30) // (limbs * n * 8)
words = polys * limbs * n
slack = (4 << 20) // 8


def run(view, label):
    bf = bi = 1e9
    for _ in range(6):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ctx.ntt_forward(view)
        e1.record()
        ctx.ntt_inverse(view)
        e2.record()
        torch.cuda.synchronize()
        bf, bi = min(bf, e0.elapsed_time(e1)), min(bi, e1.elapsed_time(e2))
    print(f"N={n} {label:48s} addr mod 2 MiB = {view.data_ptr() % (2 << 20):8d}: forward {bf:.3f} ms, inverse {bi:.3f} ms")


for trial in range(2):
    big = torch.zeros(words + slack, dtype=torch.int64, device="cuda:0")
    for off_bytes in (0, 256, 1024, 4096, 65536, 1 << 20, (1 << 20) + 4096 + 256):
        o = off_bytes // 8
        view = big[o:o + words].view(polys, limbs, n)
        view.random_(0, min(primes))
        run(view, f"allocation {trial}, offset {off_bytes} B")
    other = torch.zeros(words // 2, dtype=torch.int64, device="cuda:0")  # disturbs the allocator before the next trial
    del big
ctx.close()
