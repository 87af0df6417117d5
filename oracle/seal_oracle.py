"""ctypes binding of the CPU oracle (oracle/seal_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (cat_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libseal_oracle.so")

SCHEME_BFV = 1
SCHEME_CKKS = 2
SCHEME_BGV = 3

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


class Modulus(C.Structure):
    _fields_ = [("value", u64), ("ratio", u64 * 3), ("bit_count", C.c_int)]


class MulModOp(C.Structure):
    _fields_ = [("operand", u64), ("quotient", u64)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "seal_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libseal_oracle.so"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp = C.c_void_p
        sz = C.c_size_t
        sig = {
            "so_modulus_init": (None, [C.POINTER(Modulus), u64]),
            "so_barrett_reduce_64": (u64, [u64, C.POINTER(Modulus)]),
            "so_barrett_reduce_128": (u64, [u64p, C.POINTER(Modulus)]),
            "so_multiply_uint_mod": (u64, [u64, u64, C.POINTER(Modulus)]),
            "so_mulmod_op_set": (None, [C.POINTER(MulModOp), u64, C.POINTER(Modulus)]),
            "so_multiply_uint_mod_op": (u64, [u64, MulModOp, C.POINTER(Modulus)]),
            "so_multiply_uint_mod_lazy": (u64, [u64, MulModOp, C.POINTER(Modulus)]),
            "so_add_uint_mod": (u64, [u64, u64, C.POINTER(Modulus)]),
            "so_sub_uint_mod": (u64, [u64, u64, C.POINTER(Modulus)]),
            "so_negate_uint_mod": (u64, [u64, C.POINTER(Modulus)]),
            "so_exponentiate_uint_mod": (u64, [u64, u64, C.POINTER(Modulus)]),
            "so_try_invert_uint_mod": (C.c_int, [u64, u64, u64p]),
            "so_dot_product_mod": (u64, [u64p, u64p, sz, C.POINTER(Modulus)]),
            "so_modulo_uint": (u64, [u64p, sz, C.POINTER(Modulus)]),
            "so_is_prime": (C.c_int, [u64]),
            "so_get_primes": (C.c_int, [u64, C.c_int, sz, u64p]),
            "so_coeff_modulus_create": (C.c_int, [sz, C.POINTER(C.c_int), sz, u64p]),
            "so_bfv_default": (C.c_int, [sz, u64p, C.POINTER(sz)]),
            "so_try_minimal_primitive_root": (C.c_int, [u64, C.POINTER(Modulus), u64p]),
            "so_reverse_bits": (C.c_uint32, [C.c_uint32, C.c_int]),
            "so_ntt_tables_create": (vp, [C.c_int, u64]),
            "so_ntt_tables_destroy": (None, [vp]),
            "so_ntt_forward": (None, [u64p, vp, C.c_int]),
            "so_ntt_inverse": (None, [u64p, vp, C.c_int]),
            "so_ntt_root_power": (u64, [vp, sz]),
            "so_ntt_inv_root_power": (u64, [vp, sz]),
            "so_modulo_poly_coeffs": (None, [u64p, sz, C.POINTER(Modulus), u64p]),
            "so_add_poly_coeffmod": (None, [u64p, u64p, sz, C.POINTER(Modulus), u64p]),
            "so_sub_poly_coeffmod": (None, [u64p, u64p, sz, C.POINTER(Modulus), u64p]),
            "so_negate_poly_coeffmod": (None, [u64p, sz, C.POINTER(Modulus), u64p]),
            "so_add_poly_scalar_coeffmod": (None, [u64p, sz, u64, C.POINTER(Modulus), u64p]),
            "so_sub_poly_scalar_coeffmod": (None, [u64p, sz, u64, C.POINTER(Modulus), u64p]),
            "so_multiply_poly_scalar_coeffmod": (None, [u64p, sz, u64, C.POINTER(Modulus), u64p]),
            "so_dyadic_product_coeffmod": (None, [u64p, u64p, sz, C.POINTER(Modulus), u64p]),
            "so_galois_elt_from_step": (C.c_uint32, [C.c_int, C.c_int]),
            "so_galois_table_ntt": (None, [C.c_int, C.c_uint32, u32p]),
            "so_apply_galois_ntt": (None, [u64p, C.c_int, C.c_uint32, u64p]),
            "so_apply_galois": (None, [u64p, C.c_int, C.c_uint32, C.POINTER(Modulus), u64p]),
            "so_rnstool_create": (vp, [sz, u64p, sz, u64]),
            "so_rnstool_destroy": (None, [vp]),
            "so_rnstool_base_B_size": (sz, [vp]),
            "so_rnstool_base_Bsk_size": (sz, [vp]),
            "so_rnstool_base_Bsk": (None, [vp, u64p]),
            "so_rnstool_m_tilde": (u64, [vp]),
            "so_rnstool_inv_q_last_mod_q": (None, [vp, u64p]),
            "so_rnstool_inv_q_last_mod_t": (u64, [vp]),
            "so_mod_t_and_divide_q_last_ntt_inplace": (None, [vp, u64p, C.POINTER(vp)]),
            "so_divide_and_round_q_last_inplace": (None, [vp, u64p]),
            "so_divide_and_round_q_last_ntt_inplace": (None, [vp, u64p, C.POINTER(vp)]),
            "so_fastbconv_m_tilde": (None, [vp, u64p, u64p]),
            "so_sm_mrq": (None, [vp, u64p, u64p]),
            "so_fast_floor": (None, [vp, u64p, u64p]),
            "so_fastbconv_sk": (None, [vp, u64p, u64p]),
            "so_fast_convert_array": (None, [u64p, sz, u64p, sz, u64p, sz, u64p]),
            "so_context_create": (vp, [C.c_int, u64p, sz, C.c_int, u64]),
            "so_context_destroy": (None, [vp]),
            "so_context_tables": (vp, [vp, sz]),
            "so_context_rnstool": (vp, [vp, sz]),
            "so_ckks_multiply": (None, [vp, sz, u64p, u64p]),
            "so_ckks_multiply_general": (None, [vp, sz, u64p, sz, u64p, sz]),
            "so_bfv_multiply": (None, [vp, sz, u64p, u64p]),
            "so_ckks_square": (None, [vp, sz, u64p]),
            "so_multiply_plain_ntt": (None, [vp, sz, u64p, sz, u64p]),
            "so_add_plain_ckks": (None, [vp, sz, u64p, u64p, C.c_int]),
            "so_switch_key_inplace": (None, [vp, sz, u64p, u64p, u64p]),
            "so_relinearize": (None, [vp, sz, u64p, u64p]),
            "so_mod_switch_scale_to_next": (None, [vp, sz, u64p, sz, u64p]),
            "so_mod_switch_drop_to_next": (None, [vp, sz, u64p, sz, u64p]),
            "so_apply_galois_inplace": (None, [vp, sz, u64p, C.c_uint32, u64p]),
            "so_transform_to_ntt": (None, [vp, sz, u64p, sz]),
            "so_transform_from_ntt": (None, [vp, sz, u64p, sz]),
            "so_sample_ternary_ntt": (None, [vp, u64, u64p]),
            "so_gen_kswitch_key": (None, [vp, u64p, u64p, u64, u64p]),
            "so_gen_relin_key": (None, [vp, u64p, u64, u64p]),
            "so_gen_galois_key": (None, [vp, u64p, C.c_uint32, u64, u64p]),
            "so_encrypt_symmetric": (None, [vp, sz, u64p, u64p, u64, C.c_int, u64p]),
            "so_decrypt_phase": (None, [vp, sz, u64p, u64p, sz, C.c_int, u64p]),
            "so_fill_uniform": (None, [vp, sz, sz, u64, u64p]),
            "so_blake2b": (C.c_int, [vp, sz, vp, sz, vp, sz]),
            "so_blake2xb": (C.c_int, [vp, sz, vp, sz, vp, sz]),
            "so_prng_create": (vp, [u64p]),
            "so_prng_destroy": (None, [vp]),
            "so_prng_generate": (None, [vp, sz, vp]),
            "so_sample_poly_ternary": (None, [vp, u64p, sz, sz, u64p]),
            "so_sample_poly_cbd": (None, [vp, u64p, sz, sz, u64p]),
            "so_sample_poly_uniform": (None, [vp, u64p, sz, sz, u64p]),
            "so_seal_gen_secret_key": (None, [vp, u64p, u64p]),
            "so_seal_encrypt_zero_symmetric": (None, [vp, sz, u64p, u64p, C.c_int, u64p]),
            "so_seal_encrypt_zero_symmetric_seeded": (None, [vp, sz, u64p, u64p, C.c_int, u64p]),
            "so_seal_expand_seed": (None, [vp, sz, u64p, u64p]),
            "so_seal_encrypt_zero_asymmetric": (None, [vp, sz, u64p, u64p, C.c_int, u64p]),
            "so_seal_encrypt_zero": (None, [vp, sz, u64p, u64p, u64p, u64p]),
            "so_seal_gen_kswitch_key": (None, [vp, u64p, u64p, u64p, u64p]),
            "so_seal_gen_relin_key": (None, [vp, u64p, u64p, u64p]),
            "so_seal_gen_galois_key": (None, [vp, u64p, C.c_uint32, u64p, u64p]),
            "so_seal_dot_product_ct_sk": (None, [vp, sz, u64p, u64p, sz, C.c_int, u64p]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def ptr(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def arr(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def modulus(value: int) -> Modulus:
    m = Modulus()
    lib().so_modulus_init(C.byref(m), value)
    return m


def get_primes(factor: int, bit_size: int, count: int) -> list[int]:
    out = np.zeros(count, dtype=np.uint64)
    rc = lib().so_get_primes(factor, bit_size, count, ptr(out))
    if rc != 0:
        raise ValueError("failed to find enough qualifying primes")
    return [int(x) for x in out]


def coeff_modulus_create(n: int, bit_sizes: list[int]) -> list[int]:
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    bs = (C.c_int * len(bit_sizes))(*bit_sizes)
    rc = lib().so_coeff_modulus_create(n, bs, len(bit_sizes), ptr(out))
    if rc != 0:
        raise ValueError("bit_sizes is invalid")
    return [int(x) for x in out]


def bfv_default(n: int) -> list[int]:
    out = np.zeros(32, dtype=np.uint64)
    cnt = C.c_size_t(0)
    if lib().so_bfv_default(n, ptr(out), C.byref(cnt)) != 0:
        raise ValueError("no default coeff modulus for this degree")
    return [int(x) for x in out[: cnt.value]]


class NTTTables:
    """util/ntt.h:69-199"""

    def __init__(self, log_n: int, q: int):
        self.h = lib().so_ntt_tables_create(log_n, q)
        if not self.h:
            raise ValueError("invalid modulus")
        self.log_n, self.n, self.q = log_n, 1 << log_n, q

    def __del__(self):
        if getattr(self, "h", None):
            lib().so_ntt_tables_destroy(self.h)
            self.h = None

    def root_power(self, i):
        return int(lib().so_ntt_root_power(self.h, i))

    def inv_root_power(self, i):
        return int(lib().so_ntt_inv_root_power(self.h, i))

    def forward(self, x: np.ndarray, lazy=False) -> np.ndarray:
        x = arr(x).copy()
        assert x.size == self.n
        lib().so_ntt_forward(ptr(x), self.h, int(lazy))
        return x

    def inverse(self, x: np.ndarray, lazy=False) -> np.ndarray:
        x = arr(x).copy()
        assert x.size == self.n
        lib().so_ntt_inverse(ptr(x), self.h, int(lazy))
        return x


class RNSTool:
    """util/rns.cpp:564-787"""

    def __init__(self, n: int, q: list[int], t: int = 0):
        self.q = list(q)
        self.n = n
        qa = arr(q)
        self.h = lib().so_rnstool_create(n, ptr(qa), len(q), t)

    def __del__(self):
        if getattr(self, "h", None):
            lib().so_rnstool_destroy(self.h)
            self.h = None

    @property
    def Bsk(self) -> list[int]:
        s = lib().so_rnstool_base_Bsk_size(self.h)
        out = np.zeros(s, dtype=np.uint64)
        lib().so_rnstool_base_Bsk(self.h, ptr(out))
        return [int(x) for x in out]

    @property
    def m_tilde(self) -> int:
        return int(lib().so_rnstool_m_tilde(self.h))

    def divide_and_round_q_last_inplace(self, x):
        x = arr(x).copy()
        lib().so_divide_and_round_q_last_inplace(self.h, ptr(x))
        return x

    def divide_and_round_q_last_ntt_inplace(self, x, tables: list[NTTTables]):
        x = arr(x).copy()
        tp = (C.c_void_p * len(tables))(*[t.h for t in tables])
        lib().so_divide_and_round_q_last_ntt_inplace(self.h, ptr(x), tp)
        return x

    @property
    def inv_q_last_mod_t(self) -> int:
        return int(lib().so_rnstool_inv_q_last_mod_t(self.h))

    def mod_t_and_divide_q_last_ntt_inplace(self, x, tables: list[NTTTables]):
        x = arr(x).copy()
        tp = (C.c_void_p * len(tables))(*[t.h for t in tables])
        lib().so_mod_t_and_divide_q_last_ntt_inplace(self.h, ptr(x), tp)
        return x

    def _run(self, fn, x, out_limbs):
        x = arr(x)
        out = np.zeros(out_limbs * self.n, dtype=np.uint64)
        fn(self.h, ptr(x), ptr(out))
        return out

    def fastbconv_m_tilde(self, x):
        return self._run(lib().so_fastbconv_m_tilde, x, len(self.Bsk) + 1)

    def sm_mrq(self, x):
        return self._run(lib().so_sm_mrq, x, len(self.Bsk))

    def fast_floor(self, x):
        return self._run(lib().so_fast_floor, x, len(self.Bsk))

    def fastbconv_sk(self, x):
        return self._run(lib().so_fastbconv_sk, x, len(self.q))


class Context:
    """Key-level modulus chain + per-level tools (context.cu:607-765)."""

    def __init__(self, log_n: int, primes: list[int], scheme: int = SCHEME_CKKS, plain_modulus: int = 0):
        pa = arr(primes)
        self.h = lib().so_context_create(log_n, ptr(pa), len(primes), scheme, plain_modulus)
        if not self.h:
            raise ValueError("invalid encryption parameters")
        self.log_n, self.n, self.k = log_n, 1 << log_n, len(primes)
        self.primes = list(primes)
        self.scheme = scheme
        self.plain_modulus = plain_modulus

    def __del__(self):
        if getattr(self, "h", None):
            lib().so_context_destroy(self.h)
            self.h = None

    # -- raw-limb helpers ---------------------------------------------------------------------
    def ntt_forward(self, x: np.ndarray, limb_ids, lazy=False) -> np.ndarray:
        """x: [..., len(limb_ids), N]"""
        x = arr(x).copy()
        flat = x.reshape(-1, len(limb_ids), self.n)
        for p in range(flat.shape[0]):
            for i, li in enumerate(limb_ids):
                lib().so_ntt_forward(ptr(flat[p, i]), lib().so_context_tables(self.h, li), int(lazy))
        return x

    def ntt_inverse(self, x: np.ndarray, limb_ids, lazy=False) -> np.ndarray:
        x = arr(x).copy()
        flat = x.reshape(-1, len(limb_ids), self.n)
        for p in range(flat.shape[0]):
            for i, li in enumerate(limb_ids):
                lib().so_ntt_inverse(ptr(flat[p, i]), lib().so_context_tables(self.h, li), int(lazy))
        return x

    def uniform(self, L: int, polys: int, seed: int) -> np.ndarray:
        out = np.zeros((polys, L, self.n), dtype=np.uint64)
        lib().so_fill_uniform(self.h, L, polys, seed, ptr(out))
        return out

    # -- evaluator ops ---------------------------------------------------------------------------
    def ckks_multiply(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        L = x.shape[1]
        out = np.zeros((3, L, self.n), dtype=np.uint64)
        out[:2] = x
        y = arr(y)
        lib().so_ckks_multiply(self.h, L, ptr(out), ptr(y))
        return out

    def ckks_multiply_general(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        L = x.shape[1]
        xs, ys = x.shape[0], y.shape[0]
        out = np.zeros((xs + ys - 1, L, self.n), dtype=np.uint64)
        out[:xs] = x
        y = arr(y)
        lib().so_ckks_multiply_general(self.h, L, ptr(out), xs, ptr(y), ys)
        return out

    def ckks_square(self, x: np.ndarray) -> np.ndarray:
        L = x.shape[1]
        out = np.zeros((3, L, self.n), dtype=np.uint64)
        out[:2] = x
        lib().so_ckks_square(self.h, L, ptr(out))
        return out

    def multiply_plain_ntt(self, ct: np.ndarray, plain: np.ndarray) -> np.ndarray:
        ct, plain = arr(ct).copy(), arr(plain)
        lib().so_multiply_plain_ntt(self.h, ct.shape[1], ptr(ct), ct.shape[0], ptr(plain))
        return ct

    def add_plain(self, ct: np.ndarray, plain: np.ndarray, subtract=False) -> np.ndarray:
        ct, plain = arr(ct).copy(), arr(plain)
        lib().so_add_plain_ckks(self.h, ct.shape[1], ptr(ct), ptr(plain), int(subtract))
        return ct

    def bfv_multiply(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        L = x.shape[1]
        out = np.zeros((3, L, self.n), dtype=np.uint64)
        out[:2] = x
        y = arr(y)
        lib().so_bfv_multiply(self.h, L, ptr(out), ptr(y))
        return out

    def switch_key_inplace(self, enc: np.ndarray, target: np.ndarray, key: np.ndarray) -> np.ndarray:
        L = enc.shape[1]
        enc = arr(enc).copy()
        target, key = arr(target), arr(key)
        lib().so_switch_key_inplace(self.h, L, ptr(enc), ptr(target), ptr(key))
        return enc

    def relinearize(self, ct3: np.ndarray, key: np.ndarray) -> np.ndarray:
        L = ct3.shape[1]
        ct3 = arr(ct3).copy()
        key = arr(key)
        lib().so_relinearize(self.h, L, ptr(ct3), ptr(key))
        return ct3[:2].copy()

    def rescale(self, ct: np.ndarray) -> np.ndarray:
        size, L = ct.shape[0], ct.shape[1]
        ct = arr(ct)
        out = np.zeros((size, L - 1, self.n), dtype=np.uint64)
        lib().so_mod_switch_scale_to_next(self.h, L, ptr(ct), size, ptr(out))
        return out

    def mod_drop(self, ct: np.ndarray) -> np.ndarray:
        size, L = ct.shape[0], ct.shape[1]
        ct = arr(ct)
        out = np.zeros((size, L - 1, self.n), dtype=np.uint64)
        lib().so_mod_switch_drop_to_next(self.h, L, ptr(ct), size, ptr(out))
        return out

    def apply_galois_inplace(self, ct: np.ndarray, elt: int, key: np.ndarray) -> np.ndarray:
        L = ct.shape[1]
        ct = arr(ct).copy()
        key = arr(key)
        lib().so_apply_galois_inplace(self.h, L, ptr(ct), elt, ptr(key))
        return ct

    def transform_to_ntt(self, ct):
        ct = arr(ct).copy()
        lib().so_transform_to_ntt(self.h, ct.shape[1], ptr(ct), ct.shape[0])
        return ct

    def transform_from_ntt(self, ct):
        ct = arr(ct).copy()
        lib().so_transform_from_ntt(self.h, ct.shape[1], ptr(ct), ct.shape[0])
        return ct

    # -- key material --------------------------------------------------------------------------------
    def secret_key(self, seed: int) -> np.ndarray:
        sk = np.zeros((self.k, self.n), dtype=np.uint64)
        lib().so_sample_ternary_ntt(self.h, seed, ptr(sk))
        return sk

    def relin_key(self, sk, seed: int) -> np.ndarray:
        key = np.zeros((self.k - 1, 2, self.k, self.n), dtype=np.uint64)
        lib().so_gen_relin_key(self.h, ptr(sk), seed, ptr(key))
        return key

    def galois_key(self, sk, elt: int, seed: int) -> np.ndarray:
        key = np.zeros((self.k - 1, 2, self.k, self.n), dtype=np.uint64)
        lib().so_gen_galois_key(self.h, ptr(sk), elt, seed, ptr(key))
        return key

    def encrypt(self, L: int, sk, msg: np.ndarray, seed: int, ntt_form: bool) -> np.ndarray:
        ct = np.zeros((2, L, self.n), dtype=np.uint64)
        msg = arr(msg)
        lib().so_encrypt_symmetric(self.h, L, ptr(sk), ptr(msg), seed, int(ntt_form), ptr(ct))
        return ct

    def decrypt_phase(self, sk, ct: np.ndarray, ntt_form: bool) -> np.ndarray:
        size, L = ct.shape[0], ct.shape[1]
        ct = arr(ct)
        out = np.zeros((L, self.n), dtype=np.uint64)
        lib().so_decrypt_phase(self.h, L, ptr(sk), ptr(ct), size, int(ntt_form), ptr(out))
        return out

    # -- the reference's own randomness; seed = 8 words = a Blake2xbPRNGFactory default seed ----
    def seal_secret_key(self, seed) -> np.ndarray:
        sk = np.zeros((self.k, self.n), dtype=np.uint64)
        lib().so_seal_gen_secret_key(self.h, ptr(arr(seed)), ptr(sk))
        return sk

    def seal_public_key(self, sk, seed) -> np.ndarray:
        pk = np.zeros((2, self.k, self.n), dtype=np.uint64)
        lib().so_seal_encrypt_zero_symmetric(self.h, self.k, ptr(arr(sk)), ptr(arr(seed)), 1, ptr(pk))
        return pk

    def seal_encrypt_zero(self, L: int, seed, pk=None, sk=None) -> np.ndarray:
        ct = np.zeros((2, L, self.n), dtype=np.uint64)
        lib().so_seal_encrypt_zero(self.h, L, ptr(arr(pk)) if pk is not None else None,
                                   ptr(arr(sk)) if sk is not None else None, ptr(arr(seed)), ptr(ct))
        return ct

    def seal_encrypt_zero_symmetric(self, L: int, sk, seed, ntt_form: bool, save_seed: bool = False) -> np.ndarray:
        ct = np.zeros((2, L, self.n), dtype=np.uint64)
        fn = lib().so_seal_encrypt_zero_symmetric_seeded if save_seed else lib().so_seal_encrypt_zero_symmetric
        fn(self.h, L, ptr(arr(sk)), ptr(arr(seed)), int(ntt_form), ptr(ct))
        return ct

    def seal_expand_seed(self, L: int, seed) -> np.ndarray:
        c1 = np.zeros((L, self.n), dtype=np.uint64)
        lib().so_seal_expand_seed(self.h, L, ptr(arr(seed)), ptr(c1))
        return c1

    def seal_relin_key(self, sk, seed) -> np.ndarray:
        key = np.zeros((self.k - 1, 2, self.k, self.n), dtype=np.uint64)
        lib().so_seal_gen_relin_key(self.h, ptr(arr(sk)), ptr(arr(seed)), ptr(key))
        return key

    def seal_galois_key(self, sk, elt: int, seed) -> np.ndarray:
        key = np.zeros((self.k - 1, 2, self.k, self.n), dtype=np.uint64)
        lib().so_seal_gen_galois_key(self.h, ptr(arr(sk)), elt, ptr(arr(seed)), ptr(key))
        return key

    def seal_dot_product(self, sk, ct: np.ndarray, ntt_form: bool) -> np.ndarray:
        size, L = ct.shape[0], ct.shape[1]
        out = np.zeros((L, self.n), dtype=np.uint64)
        lib().so_seal_dot_product_ct_sk(self.h, L, ptr(arr(sk)), ptr(arr(ct)), size, int(ntt_form), ptr(out))
        return out


class PRNG:
    """Blake2xbPRNG (randomgen.cpp:201-211)"""

    def __init__(self, seed):
        self.h = lib().so_prng_create(ptr(arr(seed)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().so_prng_destroy(self.h)
            self.h = None

    def generate(self, count: int) -> bytes:
        buf = C.create_string_buffer(count)
        lib().so_prng_generate(self.h, count, buf)
        return buf.raw

    def _sample(self, fn, q, n):
        q = arr(q)
        out = np.zeros((len(q), n), dtype=np.uint64)
        fn(self.h, ptr(q), len(q), n, ptr(out))
        return out

    def ternary(self, q, n):
        return self._sample(lib().so_sample_poly_ternary, q, n)

    def cbd(self, q, n):
        return self._sample(lib().so_sample_poly_cbd, q, n)

    def uniform(self, q, n):
        return self._sample(lib().so_sample_poly_uniform, q, n)


def blake2b(data: bytes, outlen: int = 64, key: bytes = b"") -> bytes:
    out = C.create_string_buffer(outlen)
    assert lib().so_blake2b(out, outlen, data, len(data), key if key else None, len(key)) == 0
    return out.raw


def blake2xb(data: bytes, outlen: int, key: bytes = b"") -> bytes:
    out = C.create_string_buffer(outlen)
    assert lib().so_blake2xb(out, outlen, data, len(data), key if key else None, len(key)) == 0
    return out.raw
