#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations that are not bench.py's headline line (configs 3, 4, 5), one GPU,
inputs resident in HBM, each result first checked against the CPU oracle on one batch element.  Every line carries a
`roofline` object: the step's algorithmic bytes (SURVEY.md 8d formulas, below) over its measured time against 8 TB/s.
Usage: python3 tests/bench_configs.py [--steps K] [--only C]   -> one JSON object per line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import cat_amd
from cat_amd.api import to_device, to_host
from oracle import seal_oracle as so


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md


# Algorithmic bytes (compulsory traffic only, a key counted once per launch that shares it over the batch)
def bytes_key_switch(n, L, batch):
    return batch * 8 * n * (L + 2 * L + 2 * L) + 8 * n * 2 * L * (L + 1)


def bytes_tensor(n, L, batch):          # ckks / bgv / bfv multiply 2 x 2 -> 3: reads 4 L N words, writes 3 L N
    return batch * 56 * L * n


def bytes_rescale(n, L, size, batch):   # reads L, writes L - 1 limbs per polynomial
    return batch * size * (2 * L - 1) * n * 8


def bytes_galois(n, L, batch):          # the permutation of both polynomials: 16 B per coefficient
    return batch * 2 * L * n * 16


def roofline(alg_bytes, seconds, kernel):
    gbps = alg_bytes / seconds / 1e9
    return {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
            "traffic": None, "algorithmic_bytes_per_step": alg_bytes, "kernel": kernel}


def timed(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def uniform(primes, prefix, n, gen):
    out = torch.empty(tuple(prefix) + (len(primes), n), dtype=torch.int64, device="cuda")
    for j, q in enumerate(primes):
        out[..., j, :] = torch.randint(0, q, tuple(prefix) + (n,), dtype=torch.int64, device="cuda", generator=gen)
    return out


def config3(steps):
    log_n, n, t = 14, 16384, 786433
    primes = so.bfv_default(n)
    octx = so.Context(log_n, primes, so.SCHEME_BFV, t)
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_BFV, t)
    L, B = 8, 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    x, y = uniform(primes[:L], (B, 2), n, gen), uniform(primes[:L], (B, 2), n, gen)
    rk = octx.relin_key(octx.secret_key(1), 2)
    drk = to_device(rk)
    out = torch.empty((B, 3, L, n), dtype=torch.int64, device="cuda")

    def step():
        ctx.bfv_multiply(x, y, out=out)
        ctx.relinearize(out, drk)

    step()
    exp = octx.relinearize(octx.bfv_multiply(to_host(x[0]), to_host(y[0])), rk)
    ok = bool((to_host(out[0, :2]) == exp).all())
    dt = timed(step, steps)
    return {"config": 3, "workload": f"BFV N=16384, 9-prime BFVDefault (438 bit), multiply_inplace + relinearize_inplace, batch {B}",
            "ops_per_s": B / dt, "ms_per_step": dt * 1e3, "parity_checked": ok,
            "roofline": roofline(bytes_tensor(n, L, B) + bytes_key_switch(n, L, B), dt,
                                 "whole step: cat_bfv_multiply + cat_relinearize (inputs + outputs + the key once)")}


def config1(steps):
    """BASELINE config 1 (the reference's own CPU-runnable case) on the GPU: CKKS N=8192, primes {60, 40, 40, 60},
    multiply + relinearize + rescale."""
    log_n, n = 13, 8192
    primes = so.coeff_modulus_create(n, [60, 40, 40, 60])
    octx = so.Context(log_n, primes, so.SCHEME_CKKS)
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_CKKS)
    L, B = 3, 256
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    x, y = uniform(primes[:L], (B, 2), n, gen), uniform(primes[:L], (B, 2), n, gen)
    rk = octx.relin_key(octx.secret_key(1), 2)
    drk = to_device(rk)
    prod = torch.empty((B, 3, L, n), dtype=torch.int64, device="cuda")
    res = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device="cuda")

    def step():
        ctx.ckks_multiply(x, y, out=prod)
        ctx.relinearize_rescale(prod, drk, out=res)

    step()
    exp = octx.rescale(octx.relinearize(octx.ckks_multiply(to_host(x[0]), to_host(y[0])), rk))
    ok = bool((to_host(res[0]) == exp).all())
    dt = timed(step, steps)
    return {"config": 1, "workload": f"CKKS N=8192, primes 60/40/40/60, multiply + relinearize + rescale_to_next, batch {B}",
            "ops_per_s": B / dt, "ms_per_step": dt * 1e3, "parity_checked": ok,
            "roofline": roofline(bytes_tensor(n, L, B) + bytes_key_switch(n, L, B) + bytes_rescale(n, L, 2, B), dt,
                                 "whole step: cat_ckks_multiply + cat_relinearize_rescale")}


def config4(steps):
    log_n, n = 16, 65536
    primes = list(reversed(so.get_primes(2 * n, 60, 25)))
    octx = so.Context(log_n, primes, so.SCHEME_CKKS)
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_CKKS)
    B, L = 8, 24
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    ct0 = uniform(primes[:L], (B, 2), n, gen)
    gk = octx.galois_key(octx.secret_key(1), 3, 2)
    dgk = to_device(gk)
    levels = 4

    def chain():
        cur = ct0.clone()
        for _ in range(levels):
            cur = ctx.apply_galois_rescale(cur, 3, dgk)  # rotate_vector(1) + rescale_to_next, the same words
        return cur

    got = chain()
    e = to_host(ct0[0])
    for _ in range(levels):
        e = octx.rescale(octx.apply_galois_inplace(e, 3, gk))
    ok = bool((to_host(got[0]) == e).all())
    dt = timed(chain, steps)
    return {"config": 4, "workload": f"CKKS N=65536, 24+1 60-bit primes, (rotate_vector(1) + rescale_to_next) x {levels} levels from L=24, batch {B}",
            "rotate_rescale_pairs_per_s": B * levels / dt, "ms_per_chain": dt * 1e3, "parity_checked": ok,
            "roofline": roofline(sum(bytes_galois(n, l, B) + bytes_key_switch(n, l, B) + bytes_rescale(n, l, 2, B)
                                     for l in range(L, L - levels, -1)), dt,
                                 "whole chain: cat_apply_galois_rescale per level (permutation + key switch + rescale)")}


def config5(steps):
    log_n, n = 15, 32768
    primes = so.bfv_default(n)
    octx = so.Context(log_n, primes, so.SCHEME_CKKS)
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_CKKS)
    L, B = 15, 64  # 1024 pairs per GPU are processed as 16 such slices
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    x, y = uniform(primes[:L], (B, 2), n, gen), uniform(primes[:L], (B, 2), n, gen)
    rk = octx.relin_key(octx.secret_key(1), 2)
    drk = to_device(rk)
    out = torch.empty((B, 3, L, n), dtype=torch.int64, device="cuda")

    def step():
        ctx.ckks_multiply(x, y, out=out)
        ctx.relinearize(out, drk)

    step()
    exp = octx.relinearize(octx.ckks_multiply(to_host(x[0]), to_host(y[0])), rk)
    ok = bool((to_host(out[0, :2]) == exp).all())
    dt = timed(step, steps)
    return {"config": 5, "workload": f"CKKS N=32768, 16-prime BFVDefault (15 + special), multiply + relinearize, slices of {B} pairs (one GPU's share of the 8192-pair batch is 1024 pairs)",
            "ops_per_s": B / dt, "ms_per_slice": dt * 1e3, "s_per_1024_pairs": 1024 / (B / dt), "parity_checked": ok,
            "roofline": roofline(bytes_tensor(n, L, B) + bytes_key_switch(n, L, B), dt,
                                 "whole slice: cat_ckks_multiply + cat_relinearize")}


def config6(steps):
    """Not a BASELINE config: the BGV row at config 5's parameters, multiply + relinearize +
    mod_switch_to_next."""
    log_n, n, t = 15, 32768, 786433
    primes = so.bfv_default(n)
    octx = so.Context(log_n, primes, so.SCHEME_BGV, t)
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_BGV, t)
    L, B = 15, 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(6)
    x, y = uniform(primes[:L], (B, 2), n, gen), uniform(primes[:L], (B, 2), n, gen)
    rk = octx.relin_key(octx.secret_key(1), 2)
    drk = to_device(rk)
    out = torch.empty((B, 3, L, n), dtype=torch.int64, device="cuda")
    res = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device="cuda")

    two_calls = os.environ.get("CAT_BENCH_TWO_CALLS") is not None

    def step():
        ctx.bgv_multiply(x, y, out=out)
        if two_calls:
            ctx.relinearize(out, drk)
            ctx.rescale_strided(out, 2, res)
        else:
            ctx.relinearize_rescale(out, drk, out=res)  # the same words, one forward transform per limb

    step()
    exp = octx.rescale(octx.relinearize(octx.ckks_multiply(to_host(x[0]), to_host(y[0])), rk))
    ok = bool((to_host(res[0]) == exp).all())
    dt = timed(step, steps)
    return {"config": 6, "workload": f"BGV N=32768, 16-prime BFVDefault (15 + special), t=786433, multiply + relinearize + mod_switch_to_next, batch {B}",
            "ops_per_s": B / dt, "ms_per_step": dt * 1e3, "parity_checked": ok,
            "roofline": roofline(bytes_tensor(n, L, B) + bytes_key_switch(n, L, B) + bytes_rescale(n, L, 2, B), dt,
                                 "whole step: cat_bgv_multiply + cat_relinearize_rescale")}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--only", type=int, default=0, help="run one configuration (1, 3, 4, 5, or 6 = the BGV row)")
    a = ap.parse_args()
    for f in (config1, config3, config4, config5, config6):
        if a.only and f.__name__ != f"config{a.only}":
            continue
        print(json.dumps(f(a.steps)), flush=True)
        torch.cuda.empty_cache()
