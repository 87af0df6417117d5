#!/usr/bin/env python3
"""Does the second NTT pass run faster when its input was written just before (Infinity Cache)?  Forward transforms of
a 256-poly x 32-limb buffer, whole and in chunks of C polynomials (pass 2 of a chunk follows pass 1 of the same chunk);
run under `rocprofv3 --kernel-trace --stats` and compare the kernels' summed durations (launch gaps are not in them).
Usage: mall_probe.py <chunk polys, 0 = whole>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import cat_amd
from bench import get_primes_desc, uniform_limbs, LOG_N, N

chunk = int(sys.argv[1])
polys = 256
primes = get_primes_desc(2 * N, 60, 32)
ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_NONE, device=0)
gen = torch.Generator(device="cuda:0")
gen.manual_seed(1)
data = uniform_limbs(torch, primes, (polys,), N, torch.device("cuda:0"), gen)
for rep in range(3):
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    e0.record()
    if chunk == 0:
        ctx.ntt_forward(data)
    else:
        for c in range(0, polys, chunk):
            ctx.ntt_forward(data[c:c + chunk])
    e1.record()
    torch.cuda.synchronize()
    print(f"chunk {chunk}: forward {e0.elapsed_time(e1):.3f} ms wall")
