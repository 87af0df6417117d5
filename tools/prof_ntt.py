#!/usr/bin/env python3
"""Profiling driver: batched forward+inverse NTT only (BASELINE config 2 shape, fewer polys by default).
Usage: python3 tools/prof_ntt.py [polys] [reps]   (run under rocprofv3 --kernel-trace --stats / --pmc ...)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import cat_amd
from bench import get_primes_desc, uniform_limbs, LOG_N, N

polys = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
primes = get_primes_desc(2 * N, 60, 32)
ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_NONE, device=0)
gen = torch.Generator(device="cuda:0")
gen.manual_seed(1)
data = uniform_limbs(torch, primes, (polys,), N, torch.device("cuda:0"), gen)
ref = data.clone()
for _ in range(reps):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    ctx.ntt_forward(data)
    e1.record()
    ctx.ntt_inverse(data)
    e2.record()
    torch.cuda.synchronize()
    print(f"fwd {e0.elapsed_time(e1):.3f} ms  inv {e1.elapsed_time(e2):.3f} ms  "
          f"({16 * data.numel() / e0.elapsed_time(e1) / 1e6:.0f} GB/s fwd)")
assert torch.equal(data, ref)
