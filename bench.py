#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RNS polynomial arithmetic path.

Metric (BASELINE.json): CKKS N=2^16 mul+relin+rescale ops/s; batched NTT GB/s vs the HBM roofline.

  * `value`    : whole-job ops/s of multiply_inplace + relinearize_inplace + rescale_to_next on CKKS
                 ciphertexts at N=2^16 with 24 ciphertext limbs + 1 special prime (BASELINE config 4's
                 parameters), one "step" = one pass over a batch of independent ciphertext pairs that is
                 already resident in HBM.  Ranks hold independent batches (weak scaling, no data-path
                 collective; RCCL only broadcasts the relinearization key once before timing).
  * `roofline` : the dominant kernel pair of that path, the batched forward NTT, measured on BASELINE
                 config 2 (N=2^16, 32 60-bit primes, 1024 polys = 17.18 GB) with HIP events on the
                 launch stream; algorithmic bytes = 16 B/coefficient/transform.
  * `cpu_baseline`: the repo's CPU oracle (a restatement of SEAL's CPU path, kind "port") timed on
                 the host cores on a bounded sample of the same op, rank 0 at N=1 only; its output is
                 also the bit-exact check of the GPU result before any number is reported.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]   (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 16
N = 1 << LOG_N
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def get_primes_desc(factor, bits, count):
    """util::get_primes (numth.cpp:278-311) -- host-side restatement for the synthetic parameter sets."""
    def is_prime(n):
        if n < 2:
            return False
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % p == 0:
                return n == p
        d, r = n - 1, 0
        while d % 2 == 0:
            d //= 2
            r += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(r - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    out = []
    v = ((1 << bits) - 1) // factor * factor + 1
    while len(out) < count:
        if is_prime(v):
            out.append(v)
        v -= factor
    return out


def uniform_limbs(torch, primes, prefix, device, gen):
    """[*prefix, len(primes), N] int64 tensor, limb j uniform in [0, q_j)"""
    out = torch.empty(tuple(prefix) + (len(primes), N), dtype=torch.int64, device=device)
    for j, q in enumerate(primes):
        out[..., j, :] = torch.randint(0, q, tuple(prefix) + (N,), dtype=torch.int64, device=device, generator=gen)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="ciphertext pairs per step per GPU")
    ap.add_argument("--limbs", type=int, default=24, help="ciphertext limbs L (key level has L+1 primes)")
    ap.add_argument("--ntt-polys", type=int, default=1024)
    ap.add_argument("--ntt-limbs", type=int, default=32)
    ap.add_argument("--ntt-reps", type=int, default=5)
    ap.add_argument("--skip-ntt", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=10, help="CPU baseline ops per host thread")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import cat_amd
    from cat_amd.api import to_host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)

    L = args.limbs
    k = L + 1
    # CoeffModulus::Create hands out equal-size primes smallest first (modulus.cpp:177-183)
    primes = list(reversed(get_primes_desc(2 * N, 60, k)))
    ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_CKKS, device=local_rank)

    # ---- synthetic inputs, resident in HBM before the timed region -------------------------------------
    B = args.batch
    x = uniform_limbs(torch, primes[:L], (B, 2), device, gen)
    y = uniform_limbs(torch, primes[:L], (B, 2), device, gen)
    key = torch.empty((k - 1, 2, k, N), dtype=torch.int64, device=device)
    if rank == 0:
        kgen = torch.Generator(device=device)
        kgen.manual_seed(99)
        key.copy_(uniform_limbs(torch, primes, (k - 1, 2), device, kgen))
    from cat_amd import sharding

    t0 = time.perf_counter()
    sharding.broadcast_key(key, src=0)  # RCCL over xGMI: the only collective, outside the timed region
    torch.cuda.synchronize()
    key_bcast_s = time.perf_counter() - t0 if world > 1 else 0.0
    prod = torch.empty((B, 3, L, N), dtype=torch.int64, device=device)
    out = torch.empty((B, 2, L - 1, N), dtype=torch.int64, device=device)

    two_calls = os.environ.get("CAT_BENCH_TWO_CALLS") is not None  # A/B: relinearize and rescale as separate calls

    def step():
        ctx.ckks_multiply(x, y, out=prod)          # Evaluator::multiply_inplace   (evaluator.cpp:569)
        if two_calls:
            ctx.relinearize(prod, key)             # Evaluator::relinearize_inplace (evaluator.cpp:1144)
            # the reference's resize to size 2 (evaluator.cpp:1191) is a no-op on the buffer: rescale reads the
            # first two polynomials of every size-3 slot in place
            ctx.rescale_strided(prod, 2, out)      # Evaluator::rescale_to_next    (evaluator.cpp:1481)
        else:
            # the same two reference calls through the fused entry point: identical words (checked against the
            # oracle's relinearize + rescale below), one forward transform per limb instead of two
            ctx.relinearize_rescale(prod, key, out=out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, device)
    ops_per_s = world * B * args.steps / elapsed
    # algorithmic bytes per op: tensor 56LN + key switch 8N[L+2L(L+1)+4L] + rescale 2(2L-1)N8
    bytes_per_op = 56 * L * N + 8 * N * (L + 2 * L * (L + 1) + 4 * L) + 2 * (2 * L - 1) * N * 8

    result = {
        "metric": "CKKS N=2^16 mul+relin+rescale ops/sec; batched NTT GB/s vs HBM roofline",
        "value": ops_per_s,
        "unit": "ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"CKKS N=65536, L={L} ciphertext limbs + 1 special 60-bit prime: multiply + relinearize + "
                        f"rescale_to_next on {B} independent ciphertext pairs per GPU per step (BASELINE config 4 "
                        "parameters, metric op)",
            "batch_per_gpu": B,
            "calls": "cat_ckks_multiply + cat_relinearize + cat_rescale_strided" if two_calls
                     else "cat_ckks_multiply + cat_relinearize_rescale (the words of relinearize then rescale_to_next)",
            "parallelism": f"batch-sharded x{world}, no data-path collective",
            "algorithmic_bytes_per_op": bytes_per_op,
            "op_hbm_floor_frac": (ops_per_s / world) * bytes_per_op / (HBM_PEAK_GBS * 1e9),
            "key_broadcast_s": key_bcast_s,
        },
    }

    # ---- bit-exact check + CPU baseline (rank 0, N=1 only) ------------------------------------------------
    if rank == 0 and world == 1 and not args.skip_cpu:
        from oracle import seal_oracle as so

        so.build()
        octx = so.Context(LOG_N, primes, so.SCHEME_CKKS)
        hx, hy, hkey = to_host(x[0]), to_host(y[0]), to_host(key)
        threads = max(1, min(16, os.cpu_count() or 1))
        reps = args.cpu_reps  # ops per thread: ~1.5 s each on one core -> 10-20 s of wall time
        outs = [None] * threads

        def cpu_op(i):
            for _ in range(reps):
                p3 = octx.ckks_multiply(hx, hy)
                r2 = octx.relinearize(p3, hkey)
                outs[i] = octx.rescale(r2)

        t0 = time.perf_counter()
        single = octx.rescale(octx.relinearize(octx.ckks_multiply(hx, hy), hkey))  # one op on one thread
        single_s = time.perf_counter() - t0
        del single
        t0 = time.perf_counter()
        ths = [threading.Thread(target=cpu_op, args=(i,)) for i in range(threads)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        cpu_s = time.perf_counter() - t0
        ok = bool((to_host(out[0]) == outs[0]).all())
        if not ok:
            print(json.dumps({"error": "GPU result differs from the CPU oracle", **result}))
            sys.exit(1)
        result["cpu_baseline"] = {
            "value": threads * reps / cpu_s,
            "unit": "ops/s",
            "cores": threads,
            "kind": "port",
            "single_thread_ops_per_s": 1.0 / single_s,
            "sample": f"{threads * reps} x (multiply+relinearize+rescale) of the same first ciphertext pair, {reps} ops "
                      f"on each of {threads} threads (ctypes releases the GIL), {cpu_s:.1f} s wall; output equals the "
                      "GPU result bit for bit",
        }
        result["parity_checked"] = ok
        del octx

    # ---- roofline leg: batched forward/inverse NTT, BASELINE config 2 --------------------------------------
    if rank == 0 and not args.skip_ntt:
        del x, y, prod, out
        torch.cuda.empty_cache()
        nprimes = get_primes_desc(2 * N, 60, args.ntt_limbs)
        nctx = cat_amd.Context(LOG_N, nprimes, cat_amd.SCHEME_NONE, device=local_rank)
        P = args.ntt_polys
        data = uniform_limbs(torch, nprimes, (P,), device, gen)
        ref = data.clone()
        coeffs = P * args.ntt_limbs * N
        alg_bytes = 16 * coeffs
        fwd_ms, inv_ms = [], []
        for rep in range(args.ntt_reps + 1):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            nctx.ntt_forward(data)
            e1.record()
            nctx.ntt_inverse(data)
            e2.record()
            torch.cuda.synchronize()
            if rep:  # first repetition is the warm-up
                fwd_ms.append(e0.elapsed_time(e1))
                inv_ms.append(e1.elapsed_time(e2))
        roundtrip = bool(torch.equal(data, ref))
        # what a plain device-to-device copy of the same bytes reaches on this box (read + write counted)
        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ref.copy_(data)
        ce0.record()
        for _ in range(3):
            ref.copy_(data)
        ce1.record()
        torch.cuda.synchronize()
        copy_gbps = 3 * 2 * data.numel() * 8 / (ce0.elapsed_time(ce1) * 1e-3) / 1e9
        f_ms, i_ms = float(np.mean(fwd_ms)), float(np.mean(inv_ms))
        achieved = alg_bytes / (f_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "ntt_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_forward_transform")
            except Exception:
                traffic = None
        result["roofline"] = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "kernel": "batched forward NTT = ntt_fwd_cols<Wide,8,false> + ntt_fwd_rows8<true> (two launches per transform; `achieved` counts the transform's algorithmic bytes once over the pair's time)",
            "workload": f"N=65536, {args.ntt_limbs} x 60-bit primes, {P} polys ({coeffs * 8 / 1e9:.2f} GB), in place",
            "algorithmic_bytes_per_launch_pair": alg_bytes,
            "forward_ms": f_ms,
            "inverse_ms": i_ms,
            "inverse_GBps": alg_bytes / (i_ms * 1e-3) / 1e9,
            "roundtrip_exact": roundtrip,
            "copy_GBps": copy_gbps,  # measured streaming copy (bytes read + written)
            "frac_of_two_copies": (2 * alg_bytes / (f_ms * 1e-3) / 1e9) / copy_gbps,  # the transform is two passes
        }
        if not roundtrip:
            print(json.dumps({"error": "NTT round trip mismatch", **result}))
            sys.exit(1)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()  # rank 0 may still be in its roofline leg
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
