#!/usr/bin/env python3
"""Does the metric op gain from two independent batches in flight on two HIP streams (two contexts = two workspaces)?
The op alternates VALU-bound kernels (key-switch passes) with memory- or latency-bound ones (tensor product, inverse
transforms, the tail), and every kernel boundary drains the chip.  Usage: python3 tools/two_streams.py [--steps K]"""
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--limbs", type=int, default=24)
    args = ap.parse_args()
    L, k = args.limbs, args.limbs + 1
    device = torch.device("cuda:0")
    gen = torch.Generator(device=device)
    gen.manual_seed(1)
    primes = list(reversed(get_primes_desc(2 * N, 60, k)))
    key = uniform_limbs(torch, primes, (k - 1, 2), N, device, gen)

    def lane(B):
        ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_CKKS, device=0)
        x = uniform_limbs(torch, primes[:L], (B, 2), N, device, gen)
        y = uniform_limbs(torch, primes[:L], (B, 2), N, device, gen)
        prod = torch.empty((B, 3, L, N), dtype=torch.int64, device=device)
        out = torch.empty((B, 2, L - 1, N), dtype=torch.int64, device=device)

        def step():
            ctx.ckks_multiply(x, y, out=prod)
            ctx.relinearize_rescale(prod, key, out=out)

        return step

    def measure(lanes, B):
        steps = [lane(B) for _ in range(lanes)]
        streams = [torch.cuda.Stream() for _ in range(lanes)]

        def run(n):
            for _ in range(n):
                for st, s in zip(steps, streams):
                    with torch.cuda.stream(s):
                        st()

        run(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"lanes={lanes} batch/lane={B}: {lanes * B * args.steps / dt:8.1f} ops/s  "
              f"({1e3 * dt / args.steps:.3f} ms per round of {lanes * B})", flush=True)

    measure(1, 8)
    measure(2, 4)
    measure(2, 8)
    measure(4, 4)
    measure(1, 8)


if __name__ == "__main__":
    main()
