"""Bounds of the kernels' lazy arithmetic, checked in exact integer arithmetic on the CPU (no GPU, no library):
* reduce_below_2q (cat_amd/csrc/modarith.cuh): for a prime q of b = 36..60 bits and any x < 16q, the quotient estimated
  from the high word, e = mulhi32(x >> 32, floor(2^(31+b)/q)) >> (b - 33), is floor(x/q) or one less, so x - e q < 2q;
  the error term of the estimate is below 3 / 2^(b-33), so 36 bits is a conservative threshold (cat_ctx_create leaves
  red_r = 0 below it, capi.cpp, and the kernels take conditional subtractions there).
* the fused key-switch accumulator (modarith.cuh Acc / rm_mac2 / rm_fold4): with operands a < 2^62 (row-pass results
  below 2q... 4q) and key words b < 2^60 the middle sum of two digits fits 64 bits and the high word holds 63 terms."""
import random

from oracle import seal_oracle as so


def estimate(x, q):
    b = q.bit_length()
    r = (1 << (31 + b)) // q
    assert r < (1 << 32)
    return (((x >> 32) * r) >> 32) >> (b - 33)


def critical_values(q):
    for k in range(17):
        for d in (-2, -1, 0, 1, 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1):
            x = k * q + d
            if 0 <= x < 16 * q:
                yield x
    yield 16 * q - 1
    for hi in (0, 1, (16 * q - 1) >> 32):
        for lo in (0, (1 << 32) - 1):
            x = (hi << 32) | lo
            if x < 16 * q:
                yield x


def test_estimated_quotient_is_exact_or_one_less_from_36_bits():
    rng = random.Random(36)
    for bits in range(36, 61):
        primes = so.get_primes(2 * 4096, bits, 3)
        for q in primes:
            assert q.bit_length() == bits and 16 * q <= 1 << 64
            xs = list(critical_values(q)) + [rng.randrange(16 * q) for _ in range(2000)]
            for x in xs:
                e = estimate(x, q)
                assert x // q - 1 <= e <= x // q, (bits, q, x)
                assert 0 <= x - e * q < 2 * q


def test_estimated_quotient_holds_for_any_64_bit_operand():
    """Policy Wider (csrc/ntt.hip: primes of 36 to 57 bits, no guard in any of the 16 stages) ends a forward transform with
    reduce_below_2q on values up to 68q: the estimate's error does not depend on the size of x, only on b >= 36."""
    rng = random.Random(57)
    for bits in range(36, 61):
        for q in so.get_primes(2 * 4096, bits, 2):
            top = (1 << 64) - 1
            xs = [top, top - 1, (top // q) * q, (top // q) * q - 1, 68 * q - 1 if 68 * q <= top else top, 36 * q]
            xs += [rng.randrange(1 << 64) for _ in range(2000)]
            for x in xs:
                x = min(x, top)
                e = estimate(x, q)
                assert e < 1 << 32 and x // q - 1 <= e <= x // q, (bits, q, x)
                assert 0 <= x - e * q < 2 * q


def test_guard_free_forward_transform_fits_64_bits_up_to_57_bit_primes():
    """A forward butterfly adds at most 4q to the larger of its inputs (X' = u + wY < u + 4q, Y' = u + 4q - wY): sixteen
    stages on inputs below 4q (canonical data, or the key-switch raise's values modulo a prime q_J <= 4q) stay below
    68q, eight stages (a first-pass-only result) below 36q."""
    q_max = (1 << 57) - 1
    assert 4 * q_max + 16 * 4 * q_max < 1 << 64
    assert not (4 * ((1 << 58) - 1) + 16 * 4 * ((1 << 58) - 1) < 1 << 64)  # 58-bit primes would not fit: the threshold is tight


def test_reduce128_equals_the_residue_for_any_128_bit_value():
    """reduce128 (modarith.cuh, the base-conversion kernel's reduction): hi 2^64 + lo mod q as a Shoup product of hi with
    2^64 mod q plus a Barrett-64 step on lo, two conditional subtractions; restated here word for word in 64-bit
    arithmetic for primes of 20 to 61 bits (the BEHZ auxiliary primes have 61)."""
    M = (1 << 64) - 1
    rng = random.Random(128)
    for bits in (20, 30, 36, 49, 55, 60, 61):
        for q in so.get_primes(2 * 4096, bits, 2):
            ratio1 = (1 << 64) // q                      # high word of floor(2^128 / q)
            r64 = (0 - ratio1 * q) & M
            assert r64 == (1 << 64) % q
            r64_quo = (r64 << 64) // q
            cases = [(0, 0), (M, M), (M, 0), (0, M), (q - 1, q - 1), (M, q), (1, (1 << 61) - 1)]
            cases += [(rng.randrange(1 << 64), rng.randrange(1 << 64)) for _ in range(3000)]
            for lo, hi in cases:
                t1 = (r64 * hi - ((hi * r64_quo) >> 64) * q) & M
                t0 = (lo - ((lo * ratio1) >> 64) * q) & M
                assert t0 < 2 * q and t1 < 2 * q
                s = t0 + t1
                assert s < 1 << 64
                if s >= 2 * q:
                    s -= 2 * q
                if s >= q:
                    s -= q
                assert s == ((hi << 64) | lo) % q, (bits, q, lo, hi)


def test_accumulator_bounds():
    a_max, b_max = (1 << 62) - 1, (1 << 60) - 1
    a0, a1, b0, b1 = a_max & 0xffffffff, a_max >> 32, b_max & 0xffffffff, b_max >> 32
    assert 2 * (a0 * b1 + a1 * b0) < 1 << 64          # middle sum of two digits in one 64-bit word (rm_mac2)
    assert 63 * a1 * b1 + 32 * ((1 << 32) - 1) < 1 << 64  # high word: 63 terms plus the folded halves of the middle sums
    # and the whole sum fits the 128 bits Barrett-128 takes (SEAL's own bound, evaluator.cpp:2620-2707)
    q = (1 << 60) - 1
    assert 63 * (4 * q - 1) * (q - 1) < 1 << 128
