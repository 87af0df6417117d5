"""Semantic pin of the oracle's Evaluator-level restatements, in the style of the reference's own
Evaluator tests (native/tests/seal/evaluator.cpp:3354-3623: encrypt, operate, decrypt, compare):
the reference holds no deterministic known answers for these ops, so correctness of the restated
multiply / relinearize / rescale / rotate / bfv_multiply is shown by decrypting their outputs."""
from functools import reduce

import numpy as np
import pytest

LOG_N = 8
N = 1 << LOG_N


def crt_centered(res, primes):
    """res: [L][N] residues -> list of centred Python ints"""
    Q = reduce(lambda a, b: a * b, primes)
    out = []
    for j in range(res.shape[1]):
        x = 0
        for i, p in enumerate(primes):
            Qi = Q // p
            x += int(res[i, j]) * Qi * pow(Qi, -1, p)
        x %= Q
        out.append(x - Q if x > Q // 2 else x)
    return out, Q


def to_rns(vals, primes):
    return np.array([[v % p for v in vals] for p in primes], dtype=np.uint64)


def negacyclic_mul(a, b):
    n = len(a)
    out = [0] * n
    for i, x in enumerate(a):
        if x == 0:
            continue
        for j, y in enumerate(b):
            k = i + j
            if k < n:
                out[k] += x * y
            else:
                out[k - n] -= x * y
    return out


def automorphism(a, elt):
    n = len(a)
    out = [0] * n
    for i, x in enumerate(a):
        k = (i * elt) % (2 * n)
        if k < n:
            out[k] = x
        else:
            out[k - n] = -x
    return out


@pytest.fixture(scope="module")
def ckks(oracle):
    primes = oracle.coeff_modulus_create(N, [40, 30, 30, 40])
    ctx = oracle.Context(LOG_N, primes, oracle.SCHEME_CKKS)
    sk = ctx.secret_key(11)
    return ctx, primes, sk


def test_ckks_multiply_relin_rescale_decrypts(oracle, ckks):
    ctx, primes, sk = ckks
    L = 3
    rng = np.random.default_rng(5)
    delta = 1 << 30
    m1 = [int(v) for v in rng.integers(-8, 9, N)]
    m2 = [int(v) for v in rng.integers(-8, 9, N)]
    c1 = ctx.encrypt(L, sk, to_rns([v * delta for v in m1], primes[:L]), 21, True)
    c2 = ctx.encrypt(L, sk, to_rns([v * delta for v in m2], primes[:L]), 22, True)
    prod = negacyclic_mul(m1, m2)
    c3 = ctx.ckks_multiply(c1, c2)
    ph, _ = crt_centered(ctx.decrypt_phase(sk, c3, True), primes[:L])
    assert max(abs(p - delta * delta * e) for p, e in zip(ph, prod)) < delta * 2 ** 14
    rk = ctx.relin_key(sk, 31)
    c2r = ctx.relinearize(c3, rk)
    ph, _ = crt_centered(ctx.decrypt_phase(sk, c2r, True), primes[:L])
    assert max(abs(p - delta * delta * e) for p, e in zip(ph, prod)) < delta * 2 ** 14
    cr = ctx.rescale(c2r)
    ph, _ = crt_centered(ctx.decrypt_phase(sk, cr, True), primes[: L - 1])
    ql = primes[L - 1]
    assert max(abs(p - (delta * delta * e) // ql) for p, e in zip(ph, prod)) < 2 ** 15
    # general-size multiply agrees with the 2x2 fast path (evaluator.cpp:604 vs :664)
    assert (ctx.ckks_multiply_general(c1, c2) == c3).all()


def test_ckks_rotate_decrypts(oracle, ckks):
    ctx, primes, sk = ckks
    L = 3
    rng = np.random.default_rng(6)
    delta = 1 << 40
    m = [int(v) for v in rng.integers(-100, 101, N)]
    ct = ctx.encrypt(L, sk, to_rns([v * delta for v in m], primes[:L]), 23, True)
    for step in (1, -2):
        elt = oracle.lib().so_galois_elt_from_step(LOG_N, step)
        gk = ctx.galois_key(sk, elt, 41)
        rot = ctx.apply_galois_inplace(ct, elt, gk)
        ph, _ = crt_centered(ctx.decrypt_phase(sk, rot, True), primes[:L])
        exp = automorphism(m, elt)
        assert max(abs(p - delta * e) for p, e in zip(ph, exp)) < 2 ** 20
    # also at a lower level (L=2 uses key limbs {0,1,k-1})
    ct2 = ctx.mod_drop(ct)
    elt = 3
    rot = ctx.apply_galois_inplace(ct2, elt, ctx.galois_key(sk, elt, 41))
    ph, _ = crt_centered(ctx.decrypt_phase(sk, rot, True), primes[:2])
    assert max(abs(p - delta * e) for p, e in zip(ph, automorphism(m, elt))) < 2 ** 20


def test_bfv_multiply_relin_decrypts(oracle):
    t = 65537
    primes = oracle.coeff_modulus_create(N, [40, 40, 40, 41])
    ctx = oracle.Context(LOG_N, primes, oracle.SCHEME_BFV, t)
    sk = ctx.secret_key(12)
    L = 3
    q = primes[:L]
    rng = np.random.default_rng(7)
    m1 = [int(v) for v in rng.integers(0, t, N)]
    m2 = [int(v) for v in rng.integers(0, t, N)]
    Q = reduce(lambda a, b: a * b, q)
    dq = Q // t
    c1 = ctx.encrypt(L, sk, to_rns([v * dq for v in m1], q), 51, False)
    c2 = ctx.encrypt(L, sk, to_rns([v * dq for v in m2], q), 52, False)
    exp = [v % t for v in negacyclic_mul(m1, m2)]

    def decode(ct):
        ph, _ = crt_centered(ctx.decrypt_phase(sk, ct, False), q)
        return [((2 * t * p + Q) // (2 * Q)) % t for p in ph]

    assert decode(c1) == m1
    c3 = ctx.bfv_multiply(c1, c2)
    assert decode(c3) == exp
    c2r = ctx.relinearize(c3, ctx.relin_key(sk, 61))
    assert decode(c2r) == exp
    # BFV mod switch (rns.cpp:789-828) keeps the plaintext
    cs = ctx.rescale(c2r)
    ph, _ = crt_centered(ctx.decrypt_phase(sk, cs, False), q[:2])
    Q2 = q[0] * q[1]
    assert [((2 * t * p + Q2) // (2 * Q2)) % t for p in ph] == exp
    # BFV coefficient-domain Galois + key switch (evaluator.cpp:2416-2430)
    elt = 3
    rot = ctx.apply_galois_inplace(c1, elt, ctx.galois_key(sk, elt, 71))
    assert decode(rot) == [v % t for v in automorphism(m1, elt)]


def test_bgv_multiply_relin_modswitch_rotate_decrypts(oracle):
    """BGV (evaluator.cpp:710-786 multiply, :2715-2758 key-switch branch, util/rns.cpp:1203-1246 mod switch): the
    reference holds no known answers for these; decrypt-and-compare as its Evaluator tests do.  A BGV ciphertext
    decrypts as [c0 + c1 s]_q = m * correction + t e, message = that mod t times correction^-1."""
    t = 65537
    primes = oracle.coeff_modulus_create(N, [50, 40, 40, 50])
    ctx = oracle.Context(LOG_N, primes, oracle.SCHEME_BGV, t)
    sk = ctx.secret_key(13)
    L = 3
    q = primes[:L]
    rng = np.random.default_rng(8)
    m1 = [int(v) for v in rng.integers(0, t, N)]
    m2 = [int(v) for v in rng.integers(0, t, N)]
    c1 = ctx.encrypt(L, sk, to_rns(m1, q), 53, True)
    c2 = ctx.encrypt(L, sk, to_rns(m2, q), 54, True)

    def decode(ct, qs, correction=1):
        ph, _ = crt_centered(ctx.decrypt_phase(sk, ct, True), qs)
        inv = pow(correction, -1, t)
        return [(p % t) * inv % t for p in ph]

    assert decode(c1, q) == m1
    exp = [v % t for v in negacyclic_mul(m1, m2)]
    c3 = ctx.ckks_multiply(c1, c2)  # bgv_multiply is the same tensor product on NTT-form data
    assert decode(c3, q) == exp
    c2r = ctx.relinearize(c3, ctx.relin_key(sk, 62))
    assert decode(c2r, q) == exp
    # mod switch: correction factor picks up q_last^-1 mod t (evaluator.cpp:1274-1279)
    cs = ctx.rescale(c2r)
    corr = pow(q[-1], -1, t)
    assert decode(cs, q[:2], corr) == exp
    elt = 3
    rot = ctx.apply_galois_inplace(c1, elt, ctx.galois_key(sk, elt, 72))
    assert decode(rot, q) == [v % t for v in automorphism(m1, elt)]
    # RNSTool level: the constant the correction factor uses
    tool = oracle.RNSTool(N, q, t)
    assert tool.inv_q_last_mod_t == corr


def test_ntt_convolution_theorem(oracle):
    # the negacyclic NTT diagonalises multiplication mod x^N+1 (pins output semantics of a1/a2/a4)
    q = oracle.get_primes(2 * N, 60, 1)[0]
    t = oracle.NTTTables(LOG_N, q)
    rng = np.random.default_rng(8)
    a = [int(v) for v in rng.integers(0, q, N, dtype=np.uint64)]
    b = [int(v) for v in rng.integers(0, q, N, dtype=np.uint64)]
    fa, fb = t.forward(a), t.forward(b)
    prod = np.array([(int(x) * int(y)) % q for x, y in zip(fa, fb)], dtype=np.uint64)
    assert t.inverse(prod).tolist() == [v % q for v in negacyclic_mul(a, b)]
    # lazy variants stay within their documented ranges and agree mod q (ntt.h:201-270)
    lz = t.forward(a, lazy=True)
    assert (lz < 4 * q).all() and ((lz % q) == fa).all()
    li = t.inverse(fa, lazy=True)
    assert (li < 2 * q).all() and ((li % q) == np.array(a, dtype=np.uint64)).all()


def test_square_and_plain_ops_are_consistent(oracle):
    """ckks_square (evaluator.cpp:1022-1078) equals ckks_multiply(x, x); multiply_plain_ntt is the dyadic product
    per polynomial; add_plain then sub_plain is the identity."""
    n, log_n = 1024, 10
    primes = oracle.coeff_modulus_create(n, [40, 30, 30, 40])
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    x = octx.uniform(3, 2, 5)
    assert (octx.ckks_square(x) == octx.ckks_multiply(x, x)).all()
    plain = octx.uniform(3, 1, 6)[0]
    prod = octx.multiply_plain_ntt(x, plain)
    qs = [int(p) for p in primes[:3]]
    for p in range(2):
        for j in range(3):
            exp = [(int(a) * int(b)) % qs[j] for a, b in zip(x[p, j, :8], plain[j, :8])]
            assert prod[p, j, :8].tolist() == exp
    s = octx.add_plain(x, plain)
    assert (s[1] == x[1]).all() and not (s[0] == x[0]).all()
    assert (octx.add_plain(s, plain, subtract=True) == x).all()
