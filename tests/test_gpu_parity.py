"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, bit-exact (integer work: no tolerance), plus the reference's own known answers."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


@pytest.fixture(scope="module")
def cat():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import cat_amd

    return cat_amd


def dev(a):
    from cat_amd.api import to_device

    return to_device(np.ascontiguousarray(a))


def host(t):
    from cat_amd.api import to_host

    return to_host(t)


def rand_limbs(rng, primes, shape_prefix, n, mult=1):
    """uniform in [0, mult*q_j) per limb; shape prefix + [len(primes), n]"""
    out = np.empty(tuple(shape_prefix) + (len(primes), n), dtype=np.uint64)
    for j, q in enumerate(primes):
        out[..., j, :] = rng.integers(0, mult * q, size=tuple(shape_prefix) + (n,), dtype=np.uint64)
    return out


# ------------------------------------------------------------------------------------------ NTT


@pytest.mark.parametrize("log_n", list(range(1, 17)))
def test_ntt_matches_oracle_all_sizes(cat, oracle, log_n):
    n = 1 << log_n
    primes = oracle.get_primes(2 * n, 60, 2) + oracle.get_primes(2 * n, 61, 1) + oracle.get_primes(2 * n, 49, 1)
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_NONE)
    for i in range(len(primes)):
        assert ctx.root(i) == octx_root(oracle, log_n, primes[i])
    rng = np.random.default_rng(100 + log_n)
    polys = 3 if log_n <= 13 else 2
    ids = list(range(len(primes)))
    x = rand_limbs(rng, primes, (polys,), n)
    x[0, :, : min(n, 4)] = 0
    x[0, :, -1] = np.array(primes, dtype=np.uint64) - 1
    for lazy in (False, True):
        got = host(ctx.ntt_forward(dev(x), lazy=lazy))
        exp = octx.ntt_forward(x, ids, lazy=lazy)
        assert (got == exp).all(), f"forward lazy={lazy}"
        got_i = host(ctx.ntt_inverse(dev(exp if not lazy else exp % np.array(primes, dtype=np.uint64)[None, :, None]),
                                     lazy=lazy))
        exp_i = octx.ntt_inverse(exp if not lazy else exp % np.array(primes, dtype=np.uint64)[None, :, None], ids,
                                 lazy=lazy)
        assert (got_i == exp_i).all(), f"inverse lazy={lazy}"
        if not lazy:
            assert (got_i == x).all()
    # documented input ranges: forward accepts [0, 4q), inverse accepts [0, 2q)
    x4 = rand_limbs(rng, primes, (1,), n, mult=4)
    assert (host(ctx.ntt_forward(dev(x4), lazy=True)) == octx.ntt_forward(x4, ids, lazy=True)).all()
    x2 = rand_limbs(rng, primes, (1,), n, mult=2)
    assert (host(ctx.ntt_inverse(dev(x2), lazy=True)) == octx.ntt_inverse(x2, ids, lazy=True)).all()
    # limb_ids permutation / subset
    sub = [3, 1]
    xs = rand_limbs(rng, [primes[i] for i in sub], (2,), n)
    assert (host(ctx.ntt_forward(dev(xs), limb_ids=sub)) == octx.ntt_forward(xs, sub)).all()
    ctx.close()


def octx_root(oracle, log_n, q):
    import ctypes as C

    m = oracle.modulus(q)
    r = C.c_uint64(0)
    assert oracle.lib().so_try_minimal_primitive_root(2 << log_n, C.byref(m), C.byref(r)) == 1
    return r.value


@pytest.mark.parametrize("big", [61, 59])
@pytest.mark.parametrize("log_n,polys", [(12, 4), (12, 5), (13, 9), (14, 6), (16, 7)])
def test_ntt_streaming_row_pass(cat, oracle, log_n, polys, big):
    """Four or more polynomials take the poly-streaming row kernels (ntt_fwd_rows8 / ntt_inv_rows8: pairs of
    polynomials per thread, odd counts, several chunks of polynomials per row position).  With a 61-bit prime in
    the set the forward transform keeps values below 8q; with all primes below 2^60 it takes the wide-range path."""
    n = 1 << log_n
    primes = oracle.get_primes(2 * n, 60, 2) + oracle.get_primes(2 * n, big, 1) + oracle.get_primes(2 * n, 49, 1)
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_NONE)
    rng = np.random.default_rng(500 + log_n + polys)
    ids = list(range(len(primes)))
    x = rand_limbs(rng, primes, (polys,), n)
    x[0, :, :3] = 0
    x[-1, :, -1] = np.array(primes, dtype=np.uint64) - 1
    exp = octx.ntt_forward(x, ids, lazy=False)
    got = host(ctx.ntt_forward(dev(x), lazy=False))
    assert (got == exp).all()
    back = host(ctx.ntt_inverse(dev(exp), lazy=False))
    assert (back == x).all()
    assert (back == octx.ntt_inverse(exp, ids, lazy=False)).all()
    # forward input range [0, 4q) and a limb subset in permuted order
    x4 = rand_limbs(rng, primes, (polys,), n, mult=4)
    assert (host(ctx.ntt_forward(dev(x4), lazy=False)) == octx.ntt_forward(x4, ids, lazy=False)).all()
    sub = [2, 0, 3]
    xs = rand_limbs(rng, [primes[i] for i in sub], (polys,), n)
    assert (host(ctx.ntt_forward(dev(xs), limb_ids=sub)) == octx.ntt_forward(xs, sub)).all()
    assert (host(ctx.ntt_inverse(dev(xs), limb_ids=sub)) == octx.ntt_inverse(xs, sub)).all()
    ctx.close()


def test_ntt_reference_known_answers(cat):
    # tests/util/ntt.cpp:77-101 through the HIP path
    g = KA["ntt_n2"]
    ctx = cat.Context(1, [g["modulus"]], cat.SCHEME_NONE)
    for x, y in g["cases"]:
        got = host(ctx.ntt_forward(dev(np.array([[x]], dtype=np.uint64))))
        assert got.reshape(-1).tolist() == y
    ctx.close()


def test_ntt_empty_batch_is_a_noop(cat):
    import torch

    ctx = cat.Context(12, [KA["get_primes_cfg2"]["first"]], cat.SCHEME_NONE)
    x = torch.empty((0, 1, 4096), dtype=torch.int64, device="cuda")
    ctx.ntt_forward(x)
    ctx.close()


# ------------------------------------------------------------------------------- coefficient-wise


@pytest.mark.parametrize("log_n", [1, 3, 12])
def test_coefficientwise_ops_match_oracle(cat, oracle, log_n):
    import ctypes as C

    n = 1 << log_n
    primes = oracle.get_primes(2 * n, 60, 2) + oracle.get_primes(2 * n, 61, 1) + oracle.get_primes(2 * n, 30, 1)
    ctx = cat.Context(log_n, primes, cat.SCHEME_NONE)
    rng = np.random.default_rng(7)
    a = rand_limbs(rng, primes, (3,), n)
    b = rand_limbs(rng, primes, (3,), n)
    qs = np.array(primes, dtype=np.uint64)
    a[0, :, 0] = 0
    b[0, :, 0] = 0
    a[0, :, 1] = qs - 1
    b[0, :, 1] = qs - 1
    lib = oracle.lib()
    mods = [oracle.modulus(q) for q in primes]

    def per_limb(fn, *arrs, scalar=None):
        out = np.zeros_like(arrs[0])
        for p in range(out.shape[0]):
            for j in range(len(primes)):
                args = [oracle.ptr(np.ascontiguousarray(x[p, j])) for x in arrs]
                o = np.zeros(n, dtype=np.uint64)
                if scalar is None:
                    fn(*args, n, C.byref(mods[j]), oracle.ptr(o))
                else:
                    fn(*args, n, int(scalar[j]), C.byref(mods[j]), oracle.ptr(o))
                out[p, j] = o
        return out

    da, db = dev(a), dev(b)
    assert (host(ctx.add(da, db)) == per_limb(lib.so_add_poly_coeffmod, a, b)).all()
    assert (host(ctx.sub(da, db)) == per_limb(lib.so_sub_poly_coeffmod, a, b)).all()
    assert (host(ctx.negate(da)) == per_limb(lib.so_negate_poly_coeffmod, a)).all()
    assert (host(ctx.dyadic(da, db)) == per_limb(lib.so_dyadic_product_coeffmod, a, b)).all()
    # lazy inputs (< 4q) to dyadic, as bfv_multiply feeds it (evaluator.cpp:497-541)
    a4 = rand_limbs(rng, primes, (3,), n, mult=4)
    assert (host(ctx.dyadic(dev(a4), db)) == per_limb(lib.so_dyadic_product_coeffmod, a4, b)).all()
    raw = rng.integers(0, 1 << 63, size=a.shape, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    assert (host(ctx.modulo(dev(raw))) == per_limb(lib.so_modulo_poly_coeffs, raw)).all()
    sc = [int(rng.integers(0, 1 << 63)) * 2 + 1 for _ in primes]
    assert (host(ctx.mul_scalar(da, sc)) == per_limb(lib.so_multiply_poly_scalar_coeffmod, a, scalar=sc)).all()
    # Shoup scalar multiply accepts any 64-bit x (uintarithsmallmod.cu:258-266)
    assert (host(ctx.mul_scalar(dev(raw), sc)) == per_limb(lib.so_multiply_poly_scalar_coeffmod, raw, scalar=sc)).all()
    sc2 = [int(rng.integers(0, q)) for q in primes]
    assert (host(ctx.add_scalar(da, sc2)) == per_limb(lib.so_add_poly_scalar_coeffmod, a, scalar=sc2)).all()
    assert (host(ctx.sub_scalar(da, sc2)) == per_limb(lib.so_sub_poly_scalar_coeffmod, a, scalar=sc2)).all()
    with pytest.raises(ValueError):
        ctx.add_scalar(da, [q for q in primes])  # scalar >= modulus (polyarithsmallmod.cpp:150-153)
    # aliasing: result == operand1 (evaluator.cpp:652-653)
    exp = per_limb(lib.so_dyadic_product_coeffmod, a, b)
    ctx.dyadic(da, db, out=da)
    assert (host(da) == exp).all()
    ctx.close()


def test_galois_permutations(cat, oracle):
    import ctypes as C

    # reference vectors (tests/util/galois.cpp:79-113): N=8, q=17, elt=3
    ctx = cat.Context(3, [17], cat.SCHEME_NONE)
    x = dev(np.array([[KA["apply_galois"]["in"]]], dtype=np.uint64))
    assert host(ctx.apply_galois(x, 3)).reshape(-1).tolist() == KA["apply_galois"]["out"]
    assert host(ctx.apply_galois_ntt(x, 3)).reshape(-1).tolist() == KA["apply_galois_ntt"]["out"]
    with pytest.raises(ValueError):
        ctx.apply_galois_ntt(x, 3, out=x)  # galois.cpp:160-163
    with pytest.raises(ValueError):
        ctx.apply_galois_ntt(x, 4)
    ctx.close()
    log_n = 13
    n = 1 << log_n
    primes = oracle.get_primes(2 * n, 55, 3)
    ctx = cat.Context(log_n, primes, cat.SCHEME_NONE)
    rng = np.random.default_rng(9)
    a = rand_limbs(rng, primes, (2,), n)
    for elt in (3, 2 * n - 1, cat.galois_elt_from_step(log_n, -5)):
        e1, e2 = np.zeros_like(a), np.zeros_like(a)
        for p in range(2):
            for j, q in enumerate(primes):
                m = oracle.modulus(q)
                src = np.ascontiguousarray(a[p, j])
                o = np.zeros(n, dtype=np.uint64)
                oracle.lib().so_apply_galois_ntt(oracle.ptr(src), log_n, elt, oracle.ptr(o))
                e1[p, j] = o
                oracle.lib().so_apply_galois(oracle.ptr(src), log_n, elt, C.byref(m), oracle.ptr(o))
                e2[p, j] = o
        assert (host(ctx.apply_galois_ntt(dev(a), elt)) == e1).all()
        assert (host(ctx.apply_galois(dev(a), elt)) == e2).all()
    ctx.close()


# ------------------------------------------------------------------------------ evaluator level

CKKS_CASES = [
    # (log_n, bit sizes of the key-level chain)
    (12, [50, 40, 40, 50]),
    (13, [60, 40, 40, 60]),  # BASELINE config 1 parameters (4_ckks_basics upstream)
    (12, [60, 60, 60, 60, 60]),  # all-60-bit chain: special prime is the largest
]


@pytest.mark.parametrize("log_n,bits", CKKS_CASES)
def test_ckks_multiply_relinearize_rescale_chain(cat, oracle, log_n, bits):
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, bits)
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    k = len(primes)
    sk = octx.secret_key(3)
    rk = octx.relin_key(sk, 4)
    drk = dev(rk)
    batch = 3
    for L in range(k - 1, 1, -1):
        x = np.stack([octx.uniform(L, 2, 1000 + 10 * L + b) for b in range(batch)])
        y = np.stack([octx.uniform(L, 2, 2000 + 10 * L + b) for b in range(batch)])
        prod = ctx.ckks_multiply(dev(x), dev(y))
        exp3 = np.stack([octx.ckks_multiply(x[b], y[b]) for b in range(batch)])
        assert (host(prod) == exp3).all(), f"ckks_multiply L={L}"
        relin = ctx.relinearize(prod, drk)
        exp2 = np.stack([octx.relinearize(exp3[b], rk) for b in range(batch)])
        assert (host(relin.contiguous()) == exp2).all(), f"relinearize L={L}"
        res = ctx.rescale(relin.contiguous())
        expr = np.stack([octx.rescale(exp2[b]) for b in range(batch)])
        assert (host(res) == expr).all(), f"rescale L={L}"
        # in-place rescale and mod_drop
        buf = dev(exp2)
        ctx.rescale(buf, out=buf.view(-1)[: batch * 2 * (L - 1) * n].view(batch, 2, L - 1, n))
        assert (host(buf).reshape(-1)[: expr.size] == expr.reshape(-1)).all()
        drop = ctx.mod_drop(dev(exp2))
        assert (host(drop) == np.stack([octx.mod_drop(exp2[b]) for b in range(batch)])).all()
        buf = dev(exp2)
        ctx.mod_drop(buf, out=buf.view(-1)[: batch * 2 * (L - 1) * n].view(batch, 2, L - 1, n))
        assert (host(buf).reshape(-1)[: expr.size] == host(drop).reshape(-1)).all()
    ctx.close()


@pytest.mark.parametrize("log_n,bits,scheme_name", [
    (12, [50, 40, 40, 50], "ckks"),
    (13, [60, 40, 40, 60], "ckks"),
    (12, [60, 60, 60, 60, 60], "ckks"),
    (12, [60, 60, 59, 60], "ckks"),       # the largest primes a scheme context accepts, in mixed order
    (12, [30, 45, 30, 55], "ckks"),       # primes of very different sizes: both reduce branches of the round sources
    (10, [30, 27, 30], "ckks"),           # N < 4096: the two steps one after the other
    (12, [40, 40, 40, 41], "bfv"),
    (12, [50, 40, 40, 50], "bgv"),
])
def test_relinearize_rescale_fused_equals_the_two_calls(cat, oracle, log_n, bits, scheme_name):
    """cat_relinearize_rescale: one forward transform per limb serves the mod-down and the rescale correction (CKKS,
    N >= 4096); the words must equal relinearize followed by rescale_to_next / mod_switch_to_next, every level."""
    n = 1 << log_n
    scheme = {"ckks": (oracle.SCHEME_CKKS, cat.SCHEME_CKKS), "bfv": (oracle.SCHEME_BFV, cat.SCHEME_BFV),
              "bgv": (oracle.SCHEME_BGV, cat.SCHEME_BGV)}[scheme_name]
    t = 0 if scheme_name == "ckks" else 65537
    primes = oracle.coeff_modulus_create(n, bits)
    octx = oracle.Context(log_n, primes, scheme[0], t)
    ctx = cat.Context(log_n, primes, scheme[1], t)
    k = len(primes)
    rk = octx.relin_key(octx.secret_key(31), 32)
    drk = dev(rk)
    for batch in (1, 5):
        for L in range(k - 1, 1, -1):
            ct3 = np.stack([octx.uniform(L, 3, 6000 + 10 * L + b) for b in range(batch)])
            exp = np.stack([octx.rescale(octx.relinearize(ct3[b], rk)) for b in range(batch)])
            got = host(ctx.relinearize_rescale(dev(ct3), drk))
            assert (got == exp).all(), f"L={L} batch={batch}"
    with pytest.raises(ValueError):
        ctx.relinearize_rescale(dev(octx.uniform(1, 3, 1)[None]), drk)  # end of modulus switching chain reached
    # the same tail behind a rotation
    elt = cat.galois_elt_from_step(log_n, 1)
    gk = octx.galois_key(octx.secret_key(31), elt, 33)
    for L in range(k - 1, 1, -1):
        ct = np.stack([octx.uniform(L, 2, 7000 + 10 * L + b) for b in range(3)])
        exp = np.stack([octx.rescale(octx.apply_galois_inplace(ct[b], elt, gk)) for b in range(3)])
        assert (host(ctx.apply_galois_rescale(dev(ct), elt, dev(gk))) == exp).all(), f"rotate+rescale L={L}"
    ctx.close()


def test_switch_key_direct_and_rotate(cat, oracle):
    log_n = 12
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, [55, 45, 45, 45, 55])
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    sk = octx.secret_key(5)
    batch = 2
    for L in (4, 2, 1):
        enc = np.stack([octx.uniform(L, 2, 300 + b + L) for b in range(batch)])
        tgt = np.stack([octx.uniform(L, 1, 400 + b + L)[0] for b in range(batch)])
        key = octx.relin_key(sk, 6)
        got = host(ctx.switch_key(dev(enc), dev(tgt), dev(key)))
        exp = np.stack([octx.switch_key_inplace(enc[b], tgt[b], key) for b in range(batch)])
        assert (got == exp).all(), f"switch_key L={L}"
        for step in (1, -3):
            elt = cat.galois_elt_from_step(log_n, step)
            gk = octx.galois_key(sk, elt, 7)
            got = host(ctx.rotate_vector(dev(enc), step, dev(gk)))
            exp = np.stack([octx.apply_galois_inplace(enc[b], elt, gk) for b in range(batch)])
            assert (got == exp).all(), f"rotate L={L} step={step}"
    # error behaviour mirrors evaluator.cpp:2534-2573
    enc = dev(np.stack([octx.uniform(5, 2, 1)]))
    with pytest.raises(ValueError):
        ctx.switch_key(enc, dev(octx.uniform(5, 1, 2)), dev(octx.relin_key(sk, 6)))  # L == k: no such data level
    none = cat.Context(log_n, primes, cat.SCHEME_NONE)
    with pytest.raises(RuntimeError):
        none.switch_key(dev(octx.uniform(2, 2, 1)[None]), dev(octx.uniform(2, 1, 2)), dev(octx.relin_key(sk, 6)))
    none.close()
    ctx.close()


@pytest.mark.parametrize("log_n,bits,count,batch", [(12, 60, 4, 9), (13, 60, 6, 2), (12, 30, 3, 5)])
def test_switch_key_fused_kernel_prime_sizes(cat, oracle, log_n, bits, count, batch):
    """The fused row-pass + inner-product kernel at the largest and at small prime sizes, with more ciphertexts than
    XCD groups of 8 and every level."""
    n = 1 << log_n
    primes = list(reversed(oracle.get_primes(2 * n, bits, count)))
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    sk = octx.secret_key(11)
    key = octx.relin_key(sk, 12)
    dkey = dev(key)
    for L in range(count - 1, 0, -1):
        enc = np.stack([octx.uniform(L, 2, 700 + b + L) for b in range(batch)])
        tgt = np.stack([octx.uniform(L, 1, 800 + b + L)[0] for b in range(batch)])
        got = host(ctx.switch_key(dev(enc), dev(tgt), dkey))
        exp = np.stack([octx.switch_key_inplace(enc[b], tgt[b], key) for b in range(batch)])
        assert (got == exp).all(), f"switch_key L={L}"
    ctx.close()


def test_switch_key_several_ciphertexts_per_workgroup(cat, oracle):
    """With few digits and thousands of workgroups the fused key-switch kernel keeps its staged twiddles for two or four
    ciphertexts in a row (keyswitch.hip, ks_rows_mac_kernel<.., true>): 1030 ciphertexts at N = 4096, L = 3 take four per
    workgroup with a ragged last run, one 60-bit chain (policy Wide) and one 50-bit chain (Wider)."""
    log_n, n, batch = 12, 4096, 1030
    for bits in (60, 50):
        primes = list(reversed(oracle.get_primes(2 * n, bits, 4)))
        octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
        ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
        key = octx.relin_key(octx.secret_key(11), 12)
        L = 3
        enc = np.stack([octx.uniform(L, 2, 7000 + b) for b in range(batch)])
        tgt = np.stack([octx.uniform(L, 1, 9000 + b)[0] for b in range(batch)])
        got = host(ctx.switch_key(dev(enc), dev(tgt), dev(key)))
        for b in list(range(0, batch, 37)) + [batch - 3, batch - 2, batch - 1]:
            assert (got[b] == octx.switch_key_inplace(enc[b], tgt[b], key)).all(), (bits, b)
        ctx.close()


@pytest.mark.parametrize("bits,count", [(60, 8), (60, 25), (59, 25), (60, 64), (55, 64)])
def test_switch_key_extreme_operands(cat, oracle, bits, count):
    """Largest operands the lazy accumulators of the fused key-switch kernel can meet: every key word and every
    target coefficient equal to q - 1 (and 0 / alternating patterns), at the largest prime size a scheme context
    accepts (60 bits, as the reference: util/defines.h:48), with d = 7, with the metric's d = 24 and with d = 63, the
    most digits a context accepts (CAT_MAX_MODULI = 64 primes; rm_mac's carry-free high word holds 64 terms,
    modarith.cuh).  (With 61-bit primes the reference's own 128-bit accumulators wrap from d = 21 on for exactly these
    operands, evaluator.cpp:2618-2671; such contexts are rejected, test_scheme_contexts_reject_61_bit_primes.)"""
    log_n = 12
    n = 1 << log_n
    primes = list(reversed(oracle.get_primes(2 * n, bits, count)))
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    k = len(primes)
    L = k - 1
    qs = np.array(primes, dtype=np.uint64)
    key = np.broadcast_to((qs - 1)[None, None, :, None], (k - 1, 2, k, n)).copy()
    enc = np.stack([octx.uniform(L, 2, 900 + b) for b in range(3)])
    tgt = np.empty((3, L, n), dtype=np.uint64)
    tgt[0] = (qs[:L] - 1)[:, None]
    tgt[1] = 0
    tgt[2] = (qs[:L] - 1)[:, None]
    tgt[2, :, ::2] = 1
    got = host(ctx.switch_key(dev(enc), dev(tgt), dev(key)))
    exp = np.stack([octx.switch_key_inplace(enc[b], tgt[b], key) for b in range(3)])
    assert (got == exp).all()
    ctx.close()


@pytest.mark.parametrize("polys", [2, 6])
@pytest.mark.parametrize("log_n", [12, 13, 14, 16])
def test_guard_free_forward_transform_at_57_bits(cat, oracle, log_n, polys):
    """Primes of 36 to 57 bits take the forward policy without any guard (csrc/ntt.hip Wider: values grow to 68q < 2^64
    and one estimated-quotient reduction ends the transform).  The largest such primes, on the single-pass kernels
    (N = 4096, 8192), the two passes with 64- and 256-row column tiles, the 16-per-thread and the streaming row kernels:
    all-(q-1), alternating and random inputs, inputs up to 4q - 1, one 36-bit prime beside the 57-bit ones."""
    n = 1 << log_n
    primes = oracle.get_primes(2 * n, 57, 3) + oracle.get_primes(2 * n, 36, 1)
    assert max(primes).bit_length() == 57
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_NONE)
    qs = np.array(primes, dtype=np.uint64)
    ids = list(range(len(primes)))
    rng = np.random.default_rng(5700 + log_n + polys)
    x = rand_limbs(rng, primes, (polys,), n)
    x[0] = (qs - 1)[:, None]
    x[1, :, ::2] = (qs - 1)[:, None]
    x[1, :, 1::2] = 0
    exp = octx.ntt_forward(x, ids, lazy=False)
    assert (host(ctx.ntt_forward(dev(x), lazy=False)) == exp).all()
    assert (host(ctx.ntt_inverse(dev(exp), lazy=False)) == x).all()
    x4 = rand_limbs(rng, primes, (polys,), n, mult=4)
    x4[0] = (4 * qs - 1)[:, None]
    assert (host(ctx.ntt_forward(dev(x4), lazy=False)) == octx.ntt_forward(x4, ids, lazy=False)).all()
    ctx.close()


@pytest.mark.parametrize("bits", [30, 35, 36, 37, 44, 56, 57, 58, 60])
def test_wide_row_pass_final_reduction_by_prime_size(cat, oracle, bits):
    """The wide-range row passes end with an estimated-quotient reduction from 16q for primes of 36 bits and more
    (modarith.cuh reduce_below_2q) and with conditional subtractions below that: both sides of the boundary, the
    smallest and the largest size, in the streaming forward transform (all-(q-1), zero and random inputs, inputs up
    to 4q - 1) and in the fused key-switch kernel (every key word and target coefficient q - 1)."""
    log_n, n, polys = 12, 4096, 6
    primes = list(reversed(oracle.get_primes(2 * n, bits, 4)))
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    qs = np.array(primes, dtype=np.uint64)
    ids = list(range(len(primes)))
    rng = np.random.default_rng(3600 + bits)
    x = rand_limbs(rng, primes, (polys,), n)
    x[0] = (qs - 1)[:, None]
    x[1] = 0
    x[2, :, ::2] = (qs - 1)[:, None]
    assert (host(ctx.ntt_forward(dev(x), lazy=False)) == octx.ntt_forward(x, ids, lazy=False)).all()
    x4 = rand_limbs(rng, primes, (polys,), n, mult=4)
    x4[0] = (4 * qs - 1)[:, None]
    assert (host(ctx.ntt_forward(dev(x4), lazy=False)) == octx.ntt_forward(x4, ids, lazy=False)).all()
    k = len(primes)
    L = k - 1
    key = np.broadcast_to((qs - 1)[None, None, :, None], (k - 1, 2, k, n)).copy()
    enc = np.stack([octx.uniform(L, 2, 950 + b) for b in range(2)])
    tgt = np.empty((2, L, n), dtype=np.uint64)
    tgt[0] = (qs[:L] - 1)[:, None]
    tgt[1] = octx.uniform(L, 1, 960)[0]
    got = host(ctx.switch_key(dev(enc), dev(tgt), dev(key)))
    exp = np.stack([octx.switch_key_inplace(enc[b], tgt[b], key) for b in range(2)])
    assert (got == exp).all()
    ctx.close()


def test_scheme_contexts_reject_61_bit_primes(cat, oracle):
    """context.cu:202-206 (invalid_coeff_modulus_bit_count): user primes are at most 60 bits; a context without a scheme
    (transforms only) takes the 61-bit primes of Modulus."""
    n = 4096
    primes = oracle.get_primes(2 * n, 60, 2) + oracle.get_primes(2 * n, 61, 1)
    for scheme, t in ((cat.SCHEME_CKKS, 0), (cat.SCHEME_BFV, 65537), (cat.SCHEME_BGV, 65537)):
        with pytest.raises(ValueError, match="at most 60-bit"):
            cat.Context(12, primes, scheme, t)
    cat.Context(12, primes, cat.SCHEME_NONE).close()


def test_bfv_switch_key_rescale_and_galois_branches(cat, oracle):
    log_n = 12
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, [40, 40, 40, 41])
    t = 65537
    octx = oracle.Context(log_n, primes, oracle.SCHEME_BFV, t)
    ctx = cat.Context(log_n, primes, cat.SCHEME_BFV, t)
    sk = octx.secret_key(8)
    rk = octx.relin_key(sk, 9)
    batch = 2
    for L in (3, 2):
        ct3 = np.stack([octx.uniform(L, 3, 500 + b + L) for b in range(batch)])
        got = host(ctx.relinearize(dev(ct3), dev(rk)).contiguous())
        exp = np.stack([octx.relinearize(ct3[b], rk) for b in range(batch)])
        assert (got == exp).all(), f"bfv relinearize L={L}"
        res = host(ctx.rescale(dev(exp)))
        assert (res == np.stack([octx.rescale(exp[b]) for b in range(batch)])).all(), f"bfv mod switch L={L}"
        elt = 3
        gk = octx.galois_key(sk, elt, 10)
        got = host(ctx.apply_galois_inplace(dev(exp), elt, dev(gk)))
        assert (got == np.stack([octx.apply_galois_inplace(exp[b], elt, gk) for b in range(batch)])).all()
    ctx.close()


BGV_CASES = [
    # (log_n, bit sizes of the key-level chain, plain modulus)
    (12, [50, 40, 40, 50], 65537),
    (13, [60, 60, 60, 60, 60], 786433),   # all-60-bit chain: the special prime is the largest
    (10, [30, 27, 30], 12289),            # N < 4096: unfused prepare / finish kernels
    (12, [36, 36, 37], (1 << 40) + 15),   # t larger than the primes: k mod q_i reduces (rns.cpp:1228)
]


@pytest.mark.parametrize("log_n,bits,t", BGV_CASES)
def test_bgv_multiply_relinearize_modswitch_rotate(cat, oracle, log_n, bits, t):
    """the BGV branch of key switching (evaluator.cpp:2715-2758), mod_switch_to_next =
    RNSTool::mod_t_and_divide_q_last_ntt_inplace (util/rns.cpp:1203-1246), bgv_multiply (:710-841) and the
    NTT-form Galois path, every level, bit-exact against the oracle."""
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, bits)
    octx = oracle.Context(log_n, primes, oracle.SCHEME_BGV, t)
    ctx = cat.Context(log_n, primes, cat.SCHEME_BGV, t)
    k = len(primes)
    sk = octx.secret_key(21)
    rk = octx.relin_key(sk, 22)
    drk = dev(rk)
    batch = 3
    for L in range(k - 1, 0, -1):
        x = np.stack([octx.uniform(L, 2, 3000 + 10 * L + b) for b in range(batch)])
        y = np.stack([octx.uniform(L, 2, 4000 + 10 * L + b) for b in range(batch)])
        prod = ctx.bgv_multiply(dev(x), dev(y))
        exp3 = np.stack([octx.ckks_multiply(x[b], y[b]) for b in range(batch)])
        assert (host(prod) == exp3).all(), f"bgv_multiply L={L}"
        relin = ctx.relinearize(prod, drk).contiguous()
        exp2 = np.stack([octx.relinearize(exp3[b], rk) for b in range(batch)])
        assert (host(relin) == exp2).all(), f"bgv relinearize L={L}"
        tgt = np.stack([octx.uniform(L, 1, 5000 + b + L)[0] for b in range(batch)])
        got = host(ctx.switch_key(dev(x), dev(tgt), drk))
        assert (got == np.stack([octx.switch_key_inplace(x[b], tgt[b], rk) for b in range(batch)])).all()
        elt = cat.galois_elt_from_step(log_n, 1)
        gk = octx.galois_key(sk, elt, 23)
        got = host(ctx.apply_galois_inplace(dev(x), elt, dev(gk)))
        assert (got == np.stack([octx.apply_galois_inplace(x[b], elt, gk) for b in range(batch)])).all(), f"rotate L={L}"
        if L >= 2:
            res = ctx.rescale(relin)
            expr = np.stack([octx.rescale(exp2[b]) for b in range(batch)])
            assert (host(res) == expr).all(), f"bgv mod switch L={L}"
            buf = dev(exp2)
            ctx.rescale(buf, out=buf.view(-1)[: batch * 2 * (L - 1) * n].view(batch, 2, L - 1, n))
            assert (host(buf).reshape(-1)[: expr.size] == expr.reshape(-1)).all()
            # the correction-factor constant (evaluator.cpp:1274-1279)
            assert ctx.bgv_inv_q_last_mod_t(L) == pow(primes[L - 1], -1, t)
    ctx.close()
    if primes[0] < (1 << 58):
        # a prime that is not invertible modulo t: logic_error("invalid rns bases"), util/rns.cpp:780-783
        with pytest.raises(RuntimeError):
            cat.Context(log_n, primes, cat.SCHEME_BGV, 2 * primes[0])


def test_rescale_reference_known_answers(cat, oracle):
    # tests/util/rns.cpp:1013-1072 through the HIP path: q = {53, 13}, N = 2
    g = KA["divide_and_round_q_last_ntt"]
    q = g["q"]
    ctx = cat.Context(1, q, cat.SCHEME_CKKS)
    tabs = [oracle.NTTTables(1, p) for p in q]
    for xin, exp, tol in g["cases"]:
        x = np.array(xin, dtype=np.uint64)
        x[:2] = tabs[0].forward(x[:2])
        x[2:] = tabs[1].forward(x[2:])
        out = host(ctx.rescale(dev(x.reshape(1, 1, 2, 2))))
        res = tabs[0].inverse(out.reshape(-1))
        for j in range(2):
            assert (q[0] + exp[j] - int(res[j])) % q[0] <= tol
    ctx.close()


# ------------------------------------------------------------------------------------- BFV / BEHZ


@pytest.mark.parametrize("log_n,bits,t", [(3, [30, 30, 31], 17), (12, [36, 36, 37], 65537), (11, [54], 1032193),
                                          (6, [40] * 19, 65537)])
def test_behz_pieces_match_oracle(cat, oracle, log_n, bits, t):
    """fastbconv_m_tilde / sm_mrq / fast_floor / fastbconv_sk (util/rns.cpp:903-1131) against the oracle's
    RNSTool (itself pinned by tests/util/rns.cpp:460-853 in test_oracle_known_answers.py).  The conversion kernel
    handles two coefficients per thread up to 16 input limbs and one beyond: the 19-prime case walks over both."""
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, bits)
    ctx = cat.Context(log_n, primes, cat.SCHEME_BFV, t)
    top = len(primes) - 1 if len(primes) > 1 else 1
    rng = np.random.default_rng(31 + log_n)
    for L in range(1, top + 1):
        tool = oracle.RNSTool(n, primes[:L], t)
        bsk = tool.Bsk
        assert ctx.bfv_base_Bsk(L) == bsk
        polys = 3
        xq = rand_limbs(rng, primes[:L], (polys,), n)
        xq[0, :, 0] = 0
        xq[0, :, 1] = np.array(primes[:L], dtype=np.uint64) - 1
        got = host(ctx.fastbconv_m_tilde(dev(xq), L))
        exp = np.stack([tool.fastbconv_m_tilde(xq[p].reshape(-1)).reshape(len(bsk) + 1, n) for p in range(polys)])
        assert (got == exp).all(), f"fastbconv_m_tilde L={L}"
        got = host(ctx.sm_mrq(dev(exp), L))
        exp2 = np.stack([tool.sm_mrq(exp[p].reshape(-1)).reshape(len(bsk), n) for p in range(polys)])
        assert (got == exp2).all(), f"sm_mrq L={L}"
        xqb = rand_limbs(rng, primes[:L] + bsk, (polys,), n)
        got = host(ctx.fast_floor(dev(xqb), L))
        exp3 = np.stack([tool.fast_floor(xqb[p].reshape(-1)).reshape(len(bsk), n) for p in range(polys)])
        assert (got == exp3).all(), f"fast_floor L={L}"
        xb = rand_limbs(rng, bsk, (polys,), n)
        got = host(ctx.fastbconv_sk(dev(xb), L))
        exp4 = np.stack([tool.fastbconv_sk(xb[p].reshape(-1)).reshape(L, n) for p in range(polys)])
        assert (got == exp4).all(), f"fastbconv_sk L={L}"
    ctx.close()


@pytest.mark.parametrize("log_n,bits,t", [(12, [36, 36, 37], 65537), (13, [43, 43, 44, 44, 44], 786433), (10, [27], 12289),
                                          (6, [40] * 19, 65537)])
def test_bfv_multiply_relinearize_matches_oracle(cat, oracle, log_n, bits, t):
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, bits)
    octx = oracle.Context(log_n, primes, oracle.SCHEME_BFV, t)
    ctx = cat.Context(log_n, primes, cat.SCHEME_BFV, t)
    k = len(primes)
    top = k - 1 if k > 1 else 1
    batch = 2
    for L in range(top, 0, -1):
        x = np.stack([octx.uniform(L, 2, 70 + b + L) for b in range(batch)])
        y = np.stack([octx.uniform(L, 2, 80 + b + L) for b in range(batch)])
        got = ctx.bfv_multiply(dev(x), dev(y))
        exp = np.stack([octx.bfv_multiply(x[b], y[b]) for b in range(batch)])
        assert (host(got) == exp).all(), f"bfv_multiply L={L}"
        if k > 1:
            sk = octx.secret_key(11)
            rk = octx.relin_key(sk, 12)
            relin = host(ctx.relinearize(got, dev(rk)).contiguous())
            assert (relin == np.stack([octx.relinearize(exp[b], rk) for b in range(batch)])).all()
    with pytest.raises(RuntimeError):
        c2 = cat.Context(log_n, primes, cat.SCHEME_CKKS)
        try:
            c2.bfv_multiply(dev(x), dev(y))
        finally:
            c2.close()
    ctx.close()


def test_bfv_config3_multiply_relinearize(cat, oracle):
    """BASELINE config 3: BFV N=16384, BFVDefault(16384) = 9 primes / 438 bits (util/globals.cpp:52-59),
    t = 786433 (examples/6_performance.cpp:980): multiply_inplace + relinearize_inplace."""
    log_n, n, t = 14, 16384, 786433
    primes = oracle.bfv_default(n)
    assert len(primes) == 9
    octx = oracle.Context(log_n, primes, oracle.SCHEME_BFV, t)
    ctx = cat.Context(log_n, primes, cat.SCHEME_BFV, t)
    L = 8
    sk = octx.secret_key(21)
    rk = octx.relin_key(sk, 22)
    x, y = octx.uniform(L, 2, 23), octx.uniform(L, 2, 24)
    got = ctx.bfv_multiply(dev(x[None]), dev(y[None]))
    exp = octx.bfv_multiply(x, y)
    assert (host(got)[0] == exp).all()
    relin = host(ctx.relinearize(got, dev(rk)).contiguous())
    assert (relin[0] == octx.relinearize(exp, rk)).all()
    ctx.close()


def test_square_and_plaintext_operands(cat, oracle):
    """ckks_square, multiply_plain_ntt, add_plain / sub_plain on the device vs the oracle."""
    log_n = 12
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, [60, 40, 40, 60])
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    batch, L = 3, 3
    x = np.stack([octx.uniform(L, 2, 90 + b) for b in range(batch)])
    x3 = np.stack([octx.uniform(L, 3, 95 + b) for b in range(batch)])
    plains = np.stack([octx.uniform(L, 1, 99 + b)[0] for b in range(batch)])
    assert (host(ctx.ckks_square(dev(x))) == np.stack([octx.ckks_square(x[b]) for b in range(batch)])).all()
    # one plaintext per ciphertext, and one shared by the batch; size 2 and size 3
    got = host(ctx.multiply_plain_ntt(dev(x3), dev(plains)))
    assert (got == np.stack([octx.multiply_plain_ntt(x3[b], plains[b]) for b in range(batch)])).all()
    got = host(ctx.multiply_plain_ntt(dev(x), dev(plains[0])))
    assert (got == np.stack([octx.multiply_plain_ntt(x[b], plains[0]) for b in range(batch)])).all()
    dx = dev(x3)
    ctx.multiply_plain_ntt(dx, dev(plains), out=dx)  # in place
    assert (host(dx) == np.stack([octx.multiply_plain_ntt(x3[b], plains[b]) for b in range(batch)])).all()
    got = host(ctx.add_plain(dev(x3), dev(plains)))
    assert (got == np.stack([octx.add_plain(x3[b], plains[b]) for b in range(batch)])).all()
    got = host(ctx.sub_plain(dev(x), dev(plains[1])))
    assert (got == np.stack([octx.add_plain(x[b], plains[1], subtract=True) for b in range(batch)])).all()
    import torch

    big = torch.zeros((batch, 3, L, n), dtype=torch.int64, device="cuda")
    alias = big.view(-1)[: batch * 2 * L * n].view(batch, 2, L, n)
    with pytest.raises(ValueError):
        ctx.ckks_square(alias, out=big)  # destination must not alias the input
    ctx.close()


def test_python_plumbing_rejects_wrong_operands(cat, oracle):
    """cat_amd/api.py raises ValueError (not assert) on operands the C ABI cannot check: dtype, contiguity, shape of
    `out`, and the C ABI itself refuses device pointers that are not 16-byte aligned."""
    import torch

    n = 8
    primes = oracle.get_primes(2 * n, 30, 2)
    ctx = cat.Context(3, primes, cat.SCHEME_NONE)
    x = torch.zeros((3, 2, n), dtype=torch.int64, device="cuda:0")
    with pytest.raises(ValueError):
        ctx.add(x, x, out=torch.zeros((3, 1, n), dtype=torch.int64, device="cuda:0"))
    with pytest.raises(ValueError):
        ctx.add(x, x.to(torch.int32))
    with pytest.raises(ValueError):
        ctx.add(x, x.cpu())
    with pytest.raises(ValueError):
        ctx.negate(x.transpose(0, 1))
    with pytest.raises(ValueError):
        ctx.add(x[..., :4], x[..., :4])
    ctx.close()
    # an odd word offset: N = 2 polynomials are 16 bytes, a view starting one word in is 8-byte aligned only
    ctx = cat.Context(1, oracle.get_primes(4, 30, 1), cat.SCHEME_NONE)
    flat = torch.zeros(10, dtype=torch.int64, device="cuda:0")
    odd = flat[1:9].view(4, 1, 2)
    with pytest.raises(ValueError, match="16-byte aligned"):
        ctx.ntt_forward(odd)
    with pytest.raises(ValueError, match="16-byte aligned"):
        ctx.add(odd, odd)
    ctx.ntt_forward(flat[2:10].view(4, 1, 2))
    ctx.close()
