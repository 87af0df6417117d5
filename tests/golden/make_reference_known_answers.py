"""Writes tests/golden/reference_known_answers.json.

The values are the literal known-answer vectors held by the reference's own unit tests
(/root/reference/native/tests/seal/util/*.cpp), transcribed by hand as DATA (inputs and expected
outputs); each group names the file:line it comes from.  No reference code is executed or copied.
"""
import json
import os

M61 = 2305843009211596801
KA = {
    "_source": "Rayman96/CAT native/tests/seal/util/*.cpp (SEAL 4.0 unit tests), transcribed literals",
    # uintarithsmallmod.cpp:142-184  [mod, lo, hi, expected]
    "barrett_reduce_128": [
        [2, 0, 0, 0], [2, 1, 0, 1], [2, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 1],
        [3, 0, 0, 0], [3, 1, 0, 1], [3, 123, 456, 0], [3, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0],
        [13131313131313, 0, 0, 0], [13131313131313, 1, 0, 1], [13131313131313, 123, 456, 8722750765283],
        [13131313131313, 24242424242424, 79797979797979, 1010101010101],
    ],
    # uintarithsmallmod.cpp:186-212  [mod, a, b, expected]
    "multiply_uint_mod": [
        [2, 0, 0, 0], [2, 0, 1, 0], [2, 1, 0, 0], [2, 1, 1, 1],
        [10, 0, 0, 0], [10, 0, 1, 0], [10, 1, 0, 0], [10, 1, 1, 1], [10, 7, 7, 9], [10, 6, 7, 2], [10, 7, 6, 2],
        [M61, 0, 0, 0], [M61, 0, 1, 0], [M61, 1, 0, 0], [M61, 1, 1, 1],
        [M61, 1152921504605798400, 1152921504605798401, 576460752302899200],
        [M61, 1152921504605798401, 1152921504605798400, 576460752302899200],
        [M61, 1152921504605798401, 1152921504605798401, 1729382256908697601],
        [M61, 2305843009211596800, 2305843009211596800, 1],
    ],
    # uintarithsmallmod.cpp:375-404  [mod, operand, quotient]
    "mulmod_operand_quotient": [
        [3, 1, 6148914691236517205], [3, 2, 12297829382473034410],
        [2147483647, 1, 8589934596], [2147483647, 2147483646, 18446744065119617019],
        [M61, 1, 8], [M61, 2305843009211596800, 18446744073709551607],
    ],
    # uintarithsmallmod.cpp:406-444  [mod, x, y_operand, expected]
    "multiply_uint_mod_shoup": [
        [2, 0, 0, 0], [2, 1, 0, 0], [2, 0, 1, 0], [2, 1, 1, 1],
        [10, 0, 0, 0], [10, 1, 0, 0], [10, 0, 1, 0], [10, 1, 1, 1], [10, 7, 6, 2], [10, 7, 7, 9], [10, 6, 7, 2],
        [M61, 0, 0, 0], [M61, 1, 0, 0], [M61, 0, 1, 0], [M61, 1, 1, 1],
        [M61, 1152921504605798401, 1152921504605798400, 576460752302899200],
        [M61, 1152921504605798400, 1152921504605798401, 576460752302899200],
        [M61, 1152921504605798401, 1152921504605798401, 1729382256908697601],
        [M61, 2305843009211596800, 2305843009211596800, 1],
    ],
    # uintarithsmallmod.cpp:446-484  [mod, x, y_operand, expected]  (lazy: result in [0, 2q))
    "multiply_uint_mod_lazy": [
        [2, 0, 0, 0], [2, 1, 0, 0], [2, 0, 1, 0], [2, 1, 1, 1],
        [10, 7, 6, 2], [10, 7, 7, 9], [10, 6, 7, 2],
        [M61, 1152921504605798401, 1152921504605798400, 576460752302899200],
        [M61, 1152921504605798400, 1152921504605798401, 576460752302899200],
        [M61, 1152921504605798401, 1152921504605798401, 1729382256908697601],
        [M61, 2305843009211596800, 2305843009211596800, 2305843009211596802],
    ],
    # uintarithsmallmod.cpp:339-373: arrays of 2s and 3s mod 5 -> count % 5;
    # arrays of (q-1) mod q=get_prime(2048, 61) -> count
    "dot_product_mod_counts": [0, 1, 2, 15, 16, 17, 32, 64],
    # numth.cpp:201-222  [mod, degree, expected]
    "minimal_primitive_root": [[11, 2, 10], [29, 2, 28], [29, 4, 12], [1234565441, 2, 1234565440],
                               [1234565441, 8, 249725733]],
    # ntt.cpp:53-75: modulus 0xffffffffffc0001, root_powers operands
    "ntt_root_powers": {"modulus": 0xffffffffffc0001,
                        "log_n_1": [1, 288794978602139552],
                        "log_n_2": [1, 288794978602139552, 178930308976060547, 748001537669050592]},
    # ntt.cpp:77-101: N=2 forward NTT  [in, out]
    "ntt_n2": {"modulus": 0xffffffffffc0001,
               "cases": [[[0, 0], [0, 0]], [[1, 0], [1, 1]], [[1, 1], [288794978602139553, 864126526004445282]]]},
    # galois.cpp:29-42 (log_n=3)  [step, elt]
    "galois_elt_from_step": [[0, 15], [1, 3], [-3, 3], [2, 9], [-2, 9], [3, 11], [-1, 11]],
    # galois.cpp:79-113 (N=8, q=17, elt=3)
    "apply_galois": {"in": [0, 1, 2, 3, 4, 5, 6, 7], "out": [0, 14, 6, 1, 13, 7, 2, 12]},
    "apply_galois_ntt": {"in": [0, 1, 2, 3, 4, 5, 6, 7], "out": [4, 5, 7, 6, 1, 0, 2, 3]},
    # rns.cpp:404-438  [ibase, obase, in (limb-major, count 3), out]
    "fast_convert_array": [
        [[3], [2], [0, 1, 2], [0, 1, 0]],
        [[2, 3], [2], [0, 1, 0, 0, 1, 2], [0, 1, 0]],
        [[2, 3], [2, 3], [1, 1, 0, 1, 2, 2], [1, 1, 0, 1, 2, 2]],
        [[2, 3], [3, 4, 5], [0, 1, 1, 0, 1, 2], [0, 1, 2, 0, 3, 1, 0, 2, 0]],
    ],
    # rns.cpp:904-1011 divide_and_round_q_last_inplace (N=2)  exact cases [q, in, out-prefix]
    "divide_and_round_q_last": [
        [[13, 7], [0, 0, 0, 0], [0, 0]], [[13, 7], [1, 2, 1, 2], [0, 0]], [[13, 7], [12, 11, 4, 3], [4, 3]],
        [[13, 7], [6, 2, 5, 1], [3, 2]],
        [[3, 5, 7, 11], [0] * 8, [0] * 6], [[3, 5, 7, 11], [1, 2] * 4, [0] * 6],
    ],
    # rns.cpp:977-1009: approximate case, (expected - got) mod q_i <= 1
    "divide_and_round_q_last_approx": [[[3, 5, 7, 11], [0, 1, 0, 0, 4, 0, 5, 4], [2, 0, 0, 1, 5, 6]]],
    # rns.cpp:1013-1072 divide_and_round_q_last_ntt_inplace, q={53,13}, N=2; coefficient-domain
    # [in, expected, tolerance]
    "divide_and_round_q_last_ntt": {"q": [53, 13],
                                    "cases": [[[0, 0, 0, 0], [0, 0], 0], [[1, 2, 1, 2], [0, 0], 0],
                                              [[4, 12, 4, 12], [1, 2], 1], [[25, 35, 12, 9], [2, 3], 1]]},
    # get_primes(2*65536, 60, 32) first and last (numth.cpp:278-311)
    "get_primes_cfg2": {"factor": 131072, "bits": 60, "count": 32, "first": 1152921504606584833,
                        "last": 1152921504538951681},
    # polyarithsmallmod.cpp:545-600 dyadic_product_coeffmod  [mod, a, b, out]
    "dyadic_product": [[13, [1, 1, 1], [2, 3, 4], [2, 3, 4]], [13, [1, 2, 1], [2, 3, 4], [2, 6, 4]],
                       [7, [2, 1, 2], [2, 3, 4], [4, 3, 1]]],
    # polyarithsmallmod.cpp:340-375 multiply_poly_scalar_coeffmod  [mod, scalar, in, out]
    "multiply_poly_scalar": [[5, 3, [1, 3, 4], [3, 4, 2]], [5, 2, [1, 3, 4], [2, 1, 3]], [3, 2, [1, 0, 2], [2, 0, 1]]],
}

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.json")
    with open(out, "w") as f:
        json.dump(KA, f, indent=1)
    print("wrote", out)
