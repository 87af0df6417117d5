#!/usr/bin/env python3
"""Randomised differential check of the HIP path (through the C ABI) against the CPU oracle: random schemes, ring
sizes, prime chains (count and bit sizes), levels, batch sizes, Galois elements and operand ranges, every word compared.
The fixed cases of test_gpu_parity.py pick the shapes the kernels distinguish by design; this walks the space between
them.  Not collected by pytest (tests/test_gpu_fuzz.py runs a short seeded slice of it); run on a GPU box as

    python3 tests/fuzz_parity.py [--seconds S] [--seed X] [--max-log-n M]

It prints one line per case and stops at the first mismatch with the case's seed (exit status 1)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PLAIN_MODULI = [17, 12289, 65537, 786433, (1 << 20) + 7, 2, 1024]


def rand_limbs(rng, primes, prefix, n, mult=1):
    out = np.empty(tuple(prefix) + (len(primes), n), dtype=np.uint64)
    for j, q in enumerate(primes):
        out[..., j, :] = rng.integers(0, mult * q, size=tuple(prefix) + (n,), dtype=np.uint64)
    return out


def draw_case(rng, max_log_n):
    """-> dict describing one case; every choice comes from rng, so the seed reproduces it"""
    scheme = ["ckks", "ckks", "bfv", "bgv"][int(rng.integers(0, 4))]
    r = rng.random()
    if r < 0.35:
        log_n = int(rng.integers(2, 12))
    elif r < 0.70:
        log_n = 12
    elif r < 0.95:
        log_n = 13
    else:
        log_n = int(rng.integers(14, 17))  # 2^14 .. 2^16: every column-pass instantiation
    log_n = min(log_n, max_log_n)
    k = int(rng.integers(2, 7 if log_n <= 12 else 6 if log_n <= 14 else 5))
    if log_n == 12 and rng.random() < 0.12:
        k = int(rng.integers(7, 27))  # long chains: the digit loops of key switching (odd and even counts, up to 25 + 1)
    lo = max(log_n + 10, 24) if k < 7 else 40  # enough primes of each size for the chain
    style = rng.random()
    if style < 0.35:
        bits = [60] * k
    elif style < 0.60:
        # every prime of at most 57 bits: the guard-free forward policy (csrc/ntt.hip Wider), with its largest size often
        bits = [57 if rng.random() < 0.3 else int(rng.integers(max(lo, 36), 58)) for _ in range(k)]
    else:
        bits = [int(rng.integers(lo, 61)) for _ in range(k)]
    t = 0 if scheme == "ckks" else PLAIN_MODULI[int(rng.integers(0, len(PLAIN_MODULI)))]
    if scheme == "bgv" and rng.random() < 0.1:
        t = (1 << 40) + 15  # larger than some primes of the chain: k mod q_i reduces (util/rns.cpp:1228)
    batch = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 9]))
    if log_n >= 15:
        batch = min(batch, 3)  # keeps the oracle's share of a case to a second or two
    step = int(rng.choice([1, -1, 2, 3, -5, 0]))
    return dict(scheme=scheme, log_n=log_n, bits=bits, t=t, batch=batch, step=step)


def run_case(cat, so, case, seed):
    from cat_amd.api import to_device, to_host

    def dev(a):
        return to_device(np.ascontiguousarray(a))

    rng = np.random.default_rng(seed)
    log_n, bits, t, batch = case["log_n"], case["bits"], case["t"], case["batch"]
    n = 1 << log_n
    scheme = case["scheme"]
    primes = so.coeff_modulus_create(n, bits)
    oscheme = {"ckks": so.SCHEME_CKKS, "bfv": so.SCHEME_BFV, "bgv": so.SCHEME_BGV}[scheme]
    cscheme = {"ckks": cat.SCHEME_CKKS, "bfv": cat.SCHEME_BFV, "bgv": cat.SCHEME_BGV}[scheme]
    octx = so.Context(log_n, primes, oscheme, t)
    ctx = cat.Context(log_n, primes, cscheme, t)
    k = len(primes)
    checks = 0

    def same(got, exp, what):
        nonlocal checks
        checks += 1
        got = to_host(got) if not isinstance(got, np.ndarray) else got
        if got.shape != exp.shape or not (got == exp).all():
            raise AssertionError(f"{what}: mismatch")

    try:
        # transforms on a random subset of limbs, documented input ranges, both laziness modes
        ids = sorted(rng.choice(k, size=int(rng.integers(1, k + 1)), replace=False).tolist())
        sub = [primes[i] for i in ids]
        polys = int(rng.integers(1, 7))
        lazy = bool(rng.integers(0, 2))
        x = rand_limbs(rng, sub, (polys,), n, mult=4 if lazy else 1)
        exp = octx.ntt_forward(x, ids, lazy=lazy)
        same(ctx.ntt_forward(dev(x), limb_ids=ids, lazy=lazy), exp, "ntt_forward")
        y = rand_limbs(rng, sub, (polys,), n, mult=2 if lazy else 1)
        same(ctx.ntt_inverse(dev(y), limb_ids=ids, lazy=lazy), octx.ntt_inverse(y, ids, lazy=lazy), "ntt_inverse")

        # coefficient-wise ops and permutations on the same limb subset
        import ctypes as C

        lib = so.lib()
        mods = [so.modulus(q) for q in sub]
        qv = np.array(sub, dtype=np.uint64)[None, :, None]

        def per_limb(fn, *arrs, scalar=None, elt=None):
            out = np.zeros_like(arrs[0])
            for p in range(out.shape[0]):
                for j in range(len(sub)):
                    o = np.zeros(n, dtype=np.uint64)
                    args = [so.ptr(np.ascontiguousarray(v[p, j])) for v in arrs]
                    if elt is not None:
                        fn(*args, log_n, elt, *([C.byref(mods[j])] if scalar == "coeff" else []), so.ptr(o))
                    elif scalar is None:
                        fn(*args, n, C.byref(mods[j]), so.ptr(o))
                    else:
                        fn(*args, n, int(scalar[j]), C.byref(mods[j]), so.ptr(o))
                    out[p, j] = o
            return out

        u, v = rand_limbs(rng, sub, (polys,), n), rand_limbs(rng, sub, (polys,), n)
        du, dv = dev(u), dev(v)
        same(ctx.add(du, dv, limb_ids=ids), (u + v) % qv, "add")
        same(ctx.sub(du, dv, limb_ids=ids), (u + (qv - v)) % qv, "sub")
        same(ctx.negate(du, limb_ids=ids), (qv - u) % qv, "negate")
        u4 = rand_limbs(rng, sub, (polys,), n, mult=4)
        same(ctx.dyadic(dev(u4), dv, limb_ids=ids), per_limb(lib.so_dyadic_product_coeffmod, u4, v), "dyadic")
        sc = [int(rng.integers(0, 1 << 63)) * 2 + 1 for _ in sub]
        same(ctx.mul_scalar(du, sc, limb_ids=ids), per_limb(lib.so_multiply_poly_scalar_coeffmod, u, scalar=sc),
             "mul_scalar")
        gelt = 2 * int(rng.integers(0, n)) + 1
        same(ctx.apply_galois(du, gelt, limb_ids=ids), per_limb(lib.so_apply_galois, u, scalar="coeff", elt=gelt),
             f"apply_galois elt={gelt}")
        same(ctx.apply_galois_ntt(du, gelt), per_limb(lib.so_apply_galois_ntt, u, elt=gelt),
             f"apply_galois_ntt elt={gelt}")

        sk = octx.secret_key(int(rng.integers(1, 1 << 30)))
        rk = octx.relin_key(sk, int(rng.integers(1, 1 << 30)))
        drk = dev(rk)
        L = int(rng.integers(1, k))  # data level: 1 .. k-1 limbs
        s1, s2 = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30))
        a = np.stack([octx.uniform(L, 2, s1 + b) for b in range(batch)])
        b_ = np.stack([octx.uniform(L, 2, s2 + b) for b in range(batch)])
        if scheme == "bfv":
            prod = ctx.bfv_multiply(dev(a), dev(b_))
            exp3 = np.stack([octx.bfv_multiply(a[i], b_[i]) for i in range(batch)])
            same(prod, exp3, "bfv_multiply")
        elif scheme == "bgv":
            prod = ctx.bgv_multiply(dev(a), dev(b_))
            exp3 = np.stack([octx.ckks_multiply(a[i], b_[i]) for i in range(batch)])
            same(prod, exp3, "bgv_multiply")
        else:
            prod = ctx.ckks_multiply(dev(a), dev(b_))
            exp3 = np.stack([octx.ckks_multiply(a[i], b_[i]) for i in range(batch)])
            same(prod, exp3, "ckks_multiply")
            same(ctx.ckks_square(dev(a)), np.stack([octx.ckks_square(a[i]) for i in range(batch)]), "ckks_square")
            plain = octx.uniform(L, 1, s2 + 50)[0]
            same(ctx.multiply_plain_ntt(dev(a), dev(plain)),
                 np.stack([octx.multiply_plain_ntt(a[i], plain) for i in range(batch)]), "multiply_plain_ntt")
            same(ctx.add_plain(dev(a), dev(plain)), np.stack([octx.add_plain(a[i], plain) for i in range(batch)]),
                 "add_plain")
            same(ctx.sub_plain(dev(a), dev(plain)),
                 np.stack([octx.add_plain(a[i], plain, subtract=True) for i in range(batch)]), "sub_plain")
        exp2 = np.stack([octx.relinearize(exp3[i], rk) for i in range(batch)])
        same(ctx.relinearize(dev(exp3), drk).contiguous(), exp2, "relinearize")
        if L >= 2:
            expr = np.stack([octx.rescale(exp2[i]) for i in range(batch)])
            same(ctx.rescale(dev(exp2)), expr, "rescale / mod switch")
            buf = dev(exp2)  # in place: the output compacts into the front of the input
            ctx.rescale(buf, out=buf.view(-1)[: expr.size].view(*expr.shape))
            same(to_host(buf).reshape(-1)[: expr.size].reshape(expr.shape), expr, "rescale in place")
            same(ctx.relinearize_rescale(dev(exp3), drk), expr, "relinearize_rescale")
            same(ctx.mod_drop(dev(exp2)), np.stack([octx.mod_drop(exp2[i]) for i in range(batch)]), "mod_drop")
        # key switch with an arbitrary target, rotation (with its rescale for the NTT-form schemes)
        tgt = np.stack([octx.uniform(L, 1, s1 + 100 + i)[0] for i in range(batch)])
        same(ctx.switch_key(dev(a), dev(tgt), drk),
             np.stack([octx.switch_key_inplace(a[i], tgt[i], rk) for i in range(batch)]), "switch_key")
        step = case["step"] if abs(case["step"]) < n // 2 else 0  # 0 = the column swap (elt 2N - 1)
        elt = cat.galois_elt_from_step(log_n, step)
        gk = octx.galois_key(sk, elt, int(rng.integers(1, 1 << 30)))
        dgk = dev(gk)
        expg = np.stack([octx.apply_galois_inplace(a[i], elt, gk) for i in range(batch)])
        same(ctx.apply_galois_inplace(dev(a), elt, dgk), expg, f"apply_galois elt={elt}")
        if L >= 2:
            same(ctx.apply_galois_rescale(dev(a), elt, dgk), np.stack([octx.rescale(expg[i]) for i in range(batch)]),
                 f"apply_galois_rescale elt={elt}")
    finally:
        ctx.close()
    return checks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0 = by time only)")
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--max-log-n", type=int, default=16)
    args = ap.parse_args()

    import cat_amd as cat
    from oracle import seal_oracle as so

    so.build()
    t0 = time.time()
    i = total = 0
    while time.time() - t0 < args.seconds and (args.cases == 0 or i < args.cases):
        seed = args.seed + i
        case = draw_case(np.random.default_rng(seed), args.max_log_n)
        try:
            total += run_case(cat, so, case, seed)
        except Exception as e:  # noqa: BLE001 -- report the case, then fail
            print(f"FAIL seed={seed} case={case}: {type(e).__name__}: {e}", flush=True)
            return 1
        print(f"ok seed={seed} {case['scheme']} log_n={case['log_n']} bits={case['bits']} t={case['t']} "
              f"batch={case['batch']} step={case['step']}  [{time.time() - t0:.0f}s]", flush=True)
        i += 1
    print(f"fuzz: {i} cases, {total} comparisons, all bit-exact", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
