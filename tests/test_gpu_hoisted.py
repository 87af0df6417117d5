"""cat_rotate_hoisted (include/cat_amd.h): several Galois elements applied to the same ciphertexts with
the decomposition, the modulus raise and the forward transforms of the extended digits shared.

The reference has no hoisted rotation and this one cannot be bit-identical to its un-hoisted apply_galois_inplace (the
canonical digits of sigma_g(c1) are not sigma_g of the digits of c1: a negated coefficient's digit q_J - t does not
reduce to minus t's residues), so -- as the reference's own Evaluator tests do for every rotation
(native/tests/seal/evaluator.cpp, decrypt and compare) -- the check is by decryption with the ORACLE:
  * CKKS: the phase of every hoisted output equals delta * sigma_g(m) within the key-switching noise bound that
    tests/test_oracle_semantics.py uses for the oracle's own (reference-restating) rotation, 2^20;
  * BFV and BGV: the decoded plaintext equals sigma_g(m) exactly;
  * and the hoisted output agrees with the un-hoisted device result (itself bit-exact against the oracle) in every word
    of c0 + c1 s up to that noise -- i.e. the two differ by key-switching noise only.
Parity with the reference: unpinned by construction (an extension); semantics pinned as above."""
from functools import reduce

import numpy as np
import pytest

from test_oracle_semantics import automorphism, crt_centered, to_rns

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("log_n", [8, 12])
def test_ckks_hoisted_rotations_decrypt_to_the_rotated_message(cat, oracle, log_n):
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, [50, 40, 40, 40, 50])
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    sk = octx.secret_key(11)
    rng = np.random.default_rng(6)
    delta = 1 << 36
    steps = [1, -2, 5, 0]  # 0 = conjugation (elt 2N - 1)
    elts = [cat.galois_elt_from_step(log_n, s) for s in steps]
    keys = [octx.galois_key(sk, e, 40 + i) for i, e in enumerate(elts)]
    dkeys = [dev(k) for k in keys]
    prepared = [ctx.hoisted_key_prepare(k, e) for k, e in zip(dkeys, elts)]
    for L in (4, 2):
        msgs = [[int(v) for v in rng.integers(-100, 101, n)] for _ in range(3)]
        cts = np.stack([octx.encrypt(L, sk, to_rns([v * delta for v in m], primes[:L]), 20 + b, True)
                        for b, m in enumerate(msgs)])
        d_ct = dev(cts)
        out = ctx.rotate_hoisted(d_ct, elts, prepared)
        assert (host(d_ct) == cts).all()  # the input is read only
        h = host(out)
        for r, e in enumerate(elts):
            plain = host(ctx.apply_galois_inplace(d_ct.clone(), e, dkeys[r]))  # bit-exact against the oracle elsewhere
            for b, m in enumerate(msgs):
                ph, _ = crt_centered(octx.decrypt_phase(sk, h[r, b], True), primes[:L])
                exp = automorphism(m, e)
                assert max(abs(p - delta * x) for p, x in zip(ph, exp)) < 2 ** 20, (L, e, b)
                ph_u, _ = crt_centered(octx.decrypt_phase(sk, plain[b], True), primes[:L])
                assert max(abs(p - u) for p, u in zip(ph, ph_u)) < 2 ** 21
            # the part that does not go through the key switch is the same permutation in both
            if r == 0:
                assert not (h[r] == plain).all()  # ... and the switched part is indeed not the reference's words
    ctx.close()


@pytest.mark.parametrize("scheme_name", ["BFV", "BGV"])
def test_bfv_bgv_hoisted_rotations_decode_exactly(cat, oracle, scheme_name):
    log_n, n, t = 12, 4096, 65537
    primes = oracle.coeff_modulus_create(n, [40, 40, 40, 41] if scheme_name == "BFV" else [50, 40, 40, 50])
    scheme_o = getattr(oracle, "SCHEME_" + scheme_name)
    octx = oracle.Context(log_n, primes, scheme_o, t)
    ctx = cat.Context(log_n, primes, getattr(cat, "SCHEME_" + scheme_name), plain_modulus=t)
    sk = octx.secret_key(12)
    L = 3
    q = primes[:L]
    Q = reduce(lambda a, b: a * b, q)
    rng = np.random.default_rng(7)
    msgs = [[int(v) for v in rng.integers(0, t, n)] for _ in range(2)]
    if scheme_name == "BFV":
        cts = np.stack([octx.encrypt(L, sk, to_rns([v * (Q // t) for v in m], q), 51 + b, False) for b, m in enumerate(msgs)])

        def decode(ct):
            ph, _ = crt_centered(octx.decrypt_phase(sk, ct, False), q)
            return [((2 * t * p + Q) // (2 * Q)) % t for p in ph]
    else:
        cts = np.stack([octx.encrypt(L, sk, to_rns(m, q), 53 + b, True) for b, m in enumerate(msgs)])

        def decode(ct):
            ph, _ = crt_centered(octx.decrypt_phase(sk, ct, True), q)
            return [p % t for p in ph]

    elts = [3, 9, 2 * n - 1]
    dkeys = [dev(octx.galois_key(sk, e, 70 + i)) for i, e in enumerate(elts)]
    prepared = [ctx.hoisted_key_prepare(k, e) for k, e in zip(dkeys, elts)]
    out = host(ctx.rotate_hoisted(dev(cts), elts, prepared))
    for r, e in enumerate(elts):
        for b, m in enumerate(msgs):
            assert decode(out[r, b]) == [v % t for v in automorphism(m, e)], (scheme_name, e, b)
    ctx.close()


def test_hoisted_argument_checks(cat, oracle):
    import torch

    log_n, n = 10, 1024
    primes = oracle.coeff_modulus_create(n, [40, 40, 41])
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    ct = torch.zeros((1, 2, 2, n), dtype=torch.int64, device="cuda")
    key = torch.zeros((2, 2, 3, n), dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError, match="Galois element is not valid"):
        ctx.hoisted_key_prepare(key, 4)
    with pytest.raises(ValueError, match="same value as operand"):
        ctx.hoisted_key_prepare(key, 3, out=key)
    with pytest.raises(ValueError, match="Galois element is not valid"):
        ctx.rotate_hoisted(ct, [2 * n + 1], [key])
    out = torch.zeros((2, 1, 2, 2, n), dtype=torch.int64, device="cuda")
    assert ctx.rotate_hoisted(ct, [], [], out=out[:0]) is not None  # no elements: nothing to do
    nctx = cat.Context(log_n, primes[:1], cat.SCHEME_NONE)
    with pytest.raises(RuntimeError, match="keyswitching is not supported"):
        nctx.rotate_hoisted(torch.zeros((1, 2, 1, n), dtype=torch.int64, device="cuda"), [3],
                            [torch.zeros((0, 2, 1, n), dtype=torch.int64, device="cuda")])
    nctx.close()
    ctx.close()
