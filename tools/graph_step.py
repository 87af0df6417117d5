#!/usr/bin/env python3
"""Does replaying the metric step as one HIP graph (14 kernels, no host work in between) beat launching it eagerly?
Usage: python3 tools/graph_step.py [--steps K]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import cat_amd
from bench import LOG_N, N, get_primes_desc, uniform_limbs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    L, k, B = 24, 25, 8
    device = torch.device("cuda:0")
    gen = torch.Generator(device=device)
    gen.manual_seed(1)
    primes = list(reversed(get_primes_desc(2 * N, 60, k)))
    ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_CKKS, device=0)
    key = uniform_limbs(torch, primes, (k - 1, 2), N, device, gen)
    x = uniform_limbs(torch, primes[:L], (B, 2), N, device, gen)
    y = uniform_limbs(torch, primes[:L], (B, 2), N, device, gen)
    prod = torch.empty((B, 3, L, N), dtype=torch.int64, device=device)
    out = torch.empty((B, 2, L - 1, N), dtype=torch.int64, device=device)

    def step():
        ctx.ckks_multiply(x, y, out=prod)
        ctx.relinearize_rescale(prod, key, out=out)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    dt = timed(step)
    ref = out.clone()
    print(f"eager : {1e3 * dt:.3f} ms per step, {B / dt:.1f} ops/s", flush=True)
    g = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(g):
        step()
    g.replay()
    torch.cuda.synchronize()
    print("graph output equals eager output:", bool((out == ref).all()), flush=True)
    dt = timed(g.replay)
    print(f"graph : {1e3 * dt:.3f} ms per step, {B / dt:.1f} ops/s", flush=True)
    dt = timed(step)
    print(f"eager : {1e3 * dt:.3f} ms per step, {B / dt:.1f} ops/s", flush=True)


if __name__ == "__main__":
    main()
