"""CPU restatement of the plaintext side of BFV / BGV encryption and decryption, in Python
integers.  TEST INFRASTRUCTURE ONLY (see seal_oracle.h): the checker of the facade's Encryptor::encrypt and
Decryptor::decrypt for BFV and BGV.

Followed, paths relative to /root/reference/native/src/seal/:
  * util/scalingvariant.cpp:69-113  multiply_add_plain_with_scaling_variant (BFV: round(q m / t) per limb), with the
    constants of context.cu:441-491 (coeff_div_plain_modulus = floor(q / t) mod q_i, q mod t, upper half threshold);
  * encryptor.cpp:245-318           BGV plaintext lift (coefficients at or above (t + 1) / 2 stand for negatives);
  * util/rns.cpp:1133-1201          RNSTool::decrypt_scale_and_round (BFV) with fast_convert_array (:418-463) and the
    constants of RNSTool::initialize (:738-765);
  * util/rns.cpp:466-539, 1248-1252 RNSTool::decrypt_modt = BaseConverter::exact_convert_array (BGV), including its
    double-precision rounding term, summed in the reference's order;
  * decryptor.cu:103-150            bgv_decrypt: the correction factor is divided out at the end.
The reference holds no known answers for these; they are pinned by encrypt -> evaluate -> decrypt == expected
plaintext (tests/test_gpu_facade.py), as its own Evaluator tests do.
"""
from __future__ import annotations

from functools import reduce

import numpy as np


def _prod(vals):
    return reduce(lambda a, b: a * b, vals, 1)


def bfv_scale_plain(m, q, t):
    """[L][N] words added to c0 by multiply_add_plain_with_scaling_variant"""
    Q = _prod(q)
    q_mod_t = Q % t
    threshold = (t + 1) >> 1
    div = [(Q // t) % qi for qi in q]
    out = np.zeros((len(q), len(m)), dtype=np.uint64)
    for idx, v in enumerate(m):
        fix = (int(v) * q_mod_t + threshold) // t
        for i, qi in enumerate(q):
            out[i, idx] = (int(v) * div[i] + fix) % qi
    return out


def bgv_lift_plain(m, q, t):
    """[L][N] coefficient-form residues of the lifted plaintext (before the forward transform)"""
    threshold = (t + 1) >> 1
    out = np.zeros((len(q), len(m)), dtype=np.uint64)
    for idx, v in enumerate(m):
        v = int(v)
        for i, qi in enumerate(q):
            out[i, idx] = (v - t) % qi if v >= threshold else v % qi
    return out


def _fast_convert(temp_in, q, obase):
    """fast_convert_array for one coefficient: temp_in[i] canonical mod q_i"""
    Q = _prod(q)
    temp = [(int(x) * pow(Q // qi, -1, qi)) % qi for x, qi in zip(temp_in, q)]
    return [sum(ti * ((Q // qi) % p) for ti, qi in zip(temp, q)) % p for p in obase], temp


def bfv_decrypt_scale_and_round(phase, q, t, gamma):
    """phase: [L][N] coefficient form -> N plaintext coefficients mod t"""
    Q = _prod(q)
    L, n = phase.shape
    neg_inv = [(-pow(Q % p, -1, p)) % p for p in (t, gamma)]
    inv_gamma_mod_t = pow(gamma % t, -1, t)
    out = []
    for j in range(n):
        temp = [(int(phase[i, j]) * ((t * gamma) % q[i])) % q[i] for i in range(L)]
        (vt, vg), _ = _fast_convert(temp, q, (t, gamma))
        vt = (vt * neg_inv[0]) % t
        vg = (vg * neg_inv[1]) % gamma
        if vg > (gamma >> 1):
            r = (vt + (gamma - vg) % t) % t
        else:
            r = (vt - vg % t) % t
        out.append((r * inv_gamma_mod_t) % t if r else 0)
    return out


def bgv_decrypt_modt(phase, q, t, correction_factor=1):
    """phase: [L][N] coefficient form (after the inverse transform) -> N plaintext coefficients mod t"""
    Q = _prod(q)
    L, n = phase.shape
    q_mod_t = Q % t
    fix = pow(correction_factor, -1, t) if correction_factor != 1 else 1
    out = []
    for j in range(n):
        (s,), temp = _fast_convert([phase[i, j] for i in range(L)], q, (t,))
        v = 0.0
        for i in range(L):  # double_t sums in limb order, then + 0.5, then truncation (rns.cpp:516-526)
            v += float(temp[i]) / float(q[i])
        v = int(v + 0.5)
        r = (s - (v * q_mod_t) % t) % t
        out.append((r * fix) % t)
    return out
