"""Pins the CPU oracle (oracle/seal_oracle.c) against the literal known-answer vectors of the
reference's own unit tests (tests/golden/reference_known_answers.json) and against the
formula-style assertions of native/tests/seal/util/rns.cpp (BEHZ pieces)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

KA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def _op(o, operand, mod):
    y = o.MulModOp()
    o.lib().so_mulmod_op_set(C.byref(y), operand, C.byref(mod))
    return y


def test_barrett_reduce_128(oracle):
    for q, lo, hi, exp in KA["barrett_reduce_128"]:
        m = oracle.modulus(q)
        w = oracle.arr([lo, hi])
        assert oracle.lib().so_barrett_reduce_128(oracle.ptr(w), C.byref(m)) == exp


def test_multiply_uint_mod(oracle):
    for q, a, b, exp in KA["multiply_uint_mod"]:
        m = oracle.modulus(q)
        assert oracle.lib().so_multiply_uint_mod(a, b, C.byref(m)) == exp


def test_mulmod_operand(oracle):
    for q, operand, quo in KA["mulmod_operand_quotient"]:
        m = oracle.modulus(q)
        y = _op(oracle, operand, m)
        assert (y.operand, y.quotient) == (operand, quo)
    for q, x, yo, exp in KA["multiply_uint_mod_shoup"]:
        m = oracle.modulus(q)
        assert oracle.lib().so_multiply_uint_mod_op(x, _op(oracle, yo, m), C.byref(m)) == exp
    for q, x, yo, exp in KA["multiply_uint_mod_lazy"]:
        m = oracle.modulus(q)
        assert oracle.lib().so_multiply_uint_mod_lazy(x, _op(oracle, yo, m), C.byref(m)) == exp


def test_dot_product_mod(oracle):
    # uintarithsmallmod.cpp:339-373
    m = oracle.modulus(5)
    a, b = oracle.arr([2] * 64), oracle.arr([3] * 64)
    for cnt in KA["dot_product_mod_counts"]:
        assert oracle.lib().so_dot_product_mod(oracle.ptr(a), oracle.ptr(b), cnt, C.byref(m)) == cnt % 5
    q = oracle.get_primes(2048, 61, 1)[0]
    m = oracle.modulus(q)
    a = oracle.arr([q - 1] * 64)
    for cnt in KA["dot_product_mod_counts"]:
        assert oracle.lib().so_dot_product_mod(oracle.ptr(a), oracle.ptr(a), cnt, C.byref(m)) == cnt


def test_minimal_primitive_root(oracle):
    for q, degree, exp in KA["minimal_primitive_root"]:
        m = oracle.modulus(q)
        r = C.c_uint64(0)
        assert oracle.lib().so_try_minimal_primitive_root(degree, C.byref(m), C.byref(r)) == 1
        assert r.value == exp


def test_get_primes_cfg2(oracle):
    g = KA["get_primes_cfg2"]
    p = oracle.get_primes(g["factor"], g["bits"], g["count"])
    assert p[0] == g["first"] and p[-1] == g["last"]
    assert all(x % g["factor"] == 1 for x in p) and sorted(p, reverse=True) == p


def test_ntt_root_powers_and_n2(oracle):
    g = KA["ntt_root_powers"]
    for log_n, key in ((1, "log_n_1"), (2, "log_n_2")):
        t = oracle.NTTTables(log_n, g["modulus"])
        assert [t.root_power(i) for i in range(1 << log_n)] == g[key]
        if log_n == 1:  # ntt.cpp:64-66
            assert t.inv_root_power(1) == pow(t.root_power(1), -1, g["modulus"])
    t = oracle.NTTTables(1, KA["ntt_n2"]["modulus"])
    for x, y in KA["ntt_n2"]["cases"]:
        assert t.forward(x).tolist() == y


def test_ntt_roundtrip_800_polys(oracle):
    # ntt.cpp:103-133 (N=8, random round trip)
    q = 0xFFFFFFFFFFC0001
    t = oracle.NTTTables(3, q)
    rng = np.random.default_rng(1)
    assert t.inverse(np.zeros(8, dtype=np.uint64)).tolist() == [0] * 8
    for _ in range(100):
        x = rng.integers(0, q, 8, dtype=np.uint64)
        assert (t.inverse(t.forward(x)) == x).all()


def test_galois(oracle):
    for step, elt in KA["galois_elt_from_step"]:
        assert oracle.lib().so_galois_elt_from_step(3, step) == elt
    x = oracle.arr(KA["apply_galois"]["in"])
    out = np.zeros(8, dtype=np.uint64)
    m = oracle.modulus(17)
    oracle.lib().so_apply_galois(oracle.ptr(x), 3, 3, C.byref(m), oracle.ptr(out))
    assert out.tolist() == KA["apply_galois"]["out"]
    oracle.lib().so_apply_galois_ntt(oracle.ptr(x), 3, 3, oracle.ptr(out))
    assert out.tolist() == KA["apply_galois_ntt"]["out"]


def test_fast_convert_array(oracle):
    for ib, ob, xin, xout in KA["fast_convert_array"]:
        ia, oa, x = oracle.arr(ib), oracle.arr(ob), oracle.arr(xin)
        out = np.zeros(len(ob) * 3, dtype=np.uint64)
        oracle.lib().so_fast_convert_array(oracle.ptr(ia), len(ib), oracle.ptr(oa), len(ob), oracle.ptr(x), 3,
                                           oracle.ptr(out))
        assert out.tolist() == xout


def test_divide_and_round_q_last(oracle):
    for q, xin, exp in KA["divide_and_round_q_last"]:
        out = oracle.RNSTool(2, q).divide_and_round_q_last_inplace(xin)
        assert out[: len(exp)].tolist() == exp
    for q, xin, exp in KA["divide_and_round_q_last_approx"]:
        out = oracle.RNSTool(2, q).divide_and_round_q_last_inplace(xin)
        for j, e in enumerate(exp):
            qi = q[j // 2]
            assert (qi + e - int(out[j])) % qi <= 1


def test_divide_and_round_q_last_ntt(oracle):
    g = KA["divide_and_round_q_last_ntt"]
    q = g["q"]
    tabs = [oracle.NTTTables(1, p) for p in q]
    tool = oracle.RNSTool(2, q)
    for xin, exp, tol in g["cases"]:
        x = oracle.arr(xin)
        x[:2] = tabs[0].forward(x[:2])
        x[2:] = tabs[1].forward(x[2:])
        out = tool.divide_and_round_q_last_ntt_inplace(x, tabs)
        res = tabs[0].inverse(out[:2])
        for j in range(2):
            assert (q[0] + exp[j] - int(res[j])) % q[0] <= tol


def test_poly_ops(oracle):
    for q, a, b, exp in KA["dyadic_product"]:
        m = oracle.modulus(q)
        a, b = oracle.arr(a), oracle.arr(b)
        out = np.zeros(3, dtype=np.uint64)
        oracle.lib().so_dyadic_product_coeffmod(oracle.ptr(a), oracle.ptr(b), 3, C.byref(m), oracle.ptr(out))
        assert out.tolist() == exp
    for q, s, a, exp in KA["multiply_poly_scalar"]:
        m = oracle.modulus(q)
        a = oracle.arr(a)
        out = np.zeros(3, dtype=np.uint64)
        oracle.lib().so_multiply_poly_scalar_coeffmod(oracle.ptr(a), 3, s, C.byref(m), oracle.ptr(out))
        assert out.tolist() == exp


# ---- BEHZ pieces: the reference's tests state expectations as formulas (rns.cpp:460-853) ----------


def test_fastbconv_m_tilde(oracle):
    # rns.cpp:460-537
    t = oracle.RNSTool(2, [3])
    mt = t.m_tilde
    base = t.Bsk + [mt]
    assert t.fastbconv_m_tilde([0, 0]).tolist() == [0] * 6
    out = t.fastbconv_m_tilde([1, 2])
    temp, temp2 = mt % 3, (2 * mt) % 3
    assert out.tolist() == [v % b for b in base for v in (temp, temp2)]
    t = oracle.RNSTool(2, [3, 5])
    base = t.Bsk + [mt]
    out = t.fastbconv_m_tilde([1, 1, 2, 2])
    temp = ((2 * mt) % 3) * 5 + ((4 * mt) % 5) * 3
    assert out.tolist() == [temp % b for b in base for _ in range(2)]


def test_sm_mrq(oracle):
    # rns.cpp:539-672
    t = oracle.RNSTool(2, [3])
    mt = t.m_tilde
    assert t.sm_mrq([0] * 6).tolist() == [0] * 4
    assert t.sm_mrq([mt, 2 * mt, mt, 2 * mt, 0, 0]).tolist() == [1, 2, 1, 2]
    assert t.sm_mrq([3] * 6).tolist() == [0] * 4
    t = oracle.RNSTool(2, [3, 5])
    assert t.sm_mrq([mt, 2 * mt] * 3 + [0, 0]).tolist() == [1, 2] * 3
    assert t.sm_mrq([15, 30] * 4).tolist() == [0] * 6
    assert t.sm_mrq([2 * mt + 15, 2 * mt + 30] * 4).tolist() == [2] * 6


def test_fast_floor(oracle):
    # rns.cpp:674-787
    t = oracle.RNSTool(2, [3])
    assert t.fast_floor([15, 3] * 3).tolist() == [5, 1, 5, 1]
    assert t.fast_floor([17, 4] * 3).tolist() == [5, 1, 5, 1]
    t = oracle.RNSTool(2, [3, 5])
    assert t.fast_floor([15, 30] * 5).tolist() == [1, 2] * 3
    out = t.fast_floor([21, 32] * 5)
    for got, exp in zip(out.tolist(), [1, 2] * 3):
        assert abs(exp - got) <= 1


def test_fastbconv_sk(oracle):
    # rns.cpp:789-853
    t = oracle.RNSTool(2, [3])
    assert t.fastbconv_sk([0] * 4).tolist() == [0, 0]
    assert t.fastbconv_sk([1, 2, 1, 2]).tolist() == [1, 2]
    t = oracle.RNSTool(2, [3, 5])
    assert t.fastbconv_sk([1, 2] * 3).tolist() == [1, 2, 1, 2]


def test_coeff_modulus_create_order(oracle):
    # modulus.cpp:177-183: equal-bit-size primes are handed out smallest first
    p = oracle.coeff_modulus_create(8192, [60, 40, 40, 60])
    desc60 = oracle.get_primes(16384, 60, 2)
    desc40 = oracle.get_primes(16384, 40, 2)
    assert p == [desc60[1], desc40[1], desc40[0], desc60[0]]
