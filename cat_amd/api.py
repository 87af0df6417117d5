"""Python plumbing over the C ABI (include/cat_amd.h).  Tensors are torch.int64 CUDA tensors whose
bits are the uint64 words of the reference's layout [..., limb, coeff]."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

SCHEME_NONE, SCHEME_BFV, SCHEME_CKKS, SCHEME_BGV = 0, 1, 2, 3

_EXC = {1: ValueError, 2: RuntimeError, 3: IndexError, 4: RuntimeError, 5: MemoryError}


class CatError(Exception):
    pass


def _check(status: int):
    if status != 0:
        msg = _lib.load().cat_last_error().decode()
        # invalid_argument -> ValueError, logic_error -> RuntimeError, out_of_range -> IndexError
        raise _EXC.get(status, CatError)(msg)


def galois_elt_from_step(log_n: int, step: int) -> int:
    elt = _lib.load().cat_galois_elt_from_step(log_n, step)
    if elt == 0:
        raise ValueError("step count too large")
    return elt


def to_device(a: np.ndarray, device="cuda:0") -> torch.Tensor:
    if a.dtype != np.uint64:
        raise ValueError("expected an array of uint64 words")
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(device)


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


class Context:
    """One context per device (cat_ctx_create).  primes = key-level chain, last = special prime."""

    def __init__(self, log_n: int, primes, scheme: int = SCHEME_CKKS, plain_modulus: int = 0, device: int = 0):
        self.lib = _lib.load()
        self.log_n, self.n, self.k = log_n, 1 << log_n, len(primes)
        self.primes = [int(p) for p in primes]
        self.scheme = scheme
        self.device = device
        arr = (C.c_uint64 * len(primes))(*self.primes)
        h = C.c_void_p()
        _check(self.lib.cat_ctx_create(log_n, arr, len(primes), scheme, plain_modulus, device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.cat_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------------------------
    def _p(self, t: torch.Tensor, shape=None):
        """Device pointer of an operand after the checks the C ABI cannot make: int64 words, contiguous, on this
        context's device, and (for outputs and fixed-shape operands) exactly the expected shape."""
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
            raise ValueError("operands are contiguous torch.int64 CUDA tensors")
        if t.device.index != self.device:
            raise ValueError(f"operand lives on cuda:{t.device.index}, the context on cuda:{self.device}")
        if t.shape[-1] != self.n:
            raise ValueError(f"the last dimension must be the polynomial degree {self.n}")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return C.c_void_p(t.data_ptr())

    def _key(self, key: torch.Tensor):
        """one KSwitchKeys entry: [k-1, 2, k, N] words"""
        return self._p(key, (self.k - 1, 2, self.k, self.n))

    def _stream(self):
        # the stream of the CONTEXT's device (every C call makes that device current)
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ids(limb_ids):
        if limb_ids is None:
            return None
        return (C.c_uint32 * len(limb_ids))(*[int(i) for i in limb_ids])

    def root(self, i: int) -> int:
        return int(self.lib.cat_ctx_root(self.h, i))

    def moduli(self):
        return [int(self.lib.cat_ctx_modulus(self.h, i)) for i in range(self.lib.cat_ctx_n_moduli(self.h))]

    def reserve(self, nbytes: int):
        _check(self.lib.cat_ctx_reserve(self.h, nbytes))

    def workspace_need(self, batch: int) -> int:
        return int(self.lib.cat_ctx_workspace_need(self.h, batch))

    def workspace_bytes(self) -> int:
        return int(self.lib.cat_ctx_workspace_bytes(self.h))

    def workspace_grew(self) -> int:
        return int(self.lib.cat_ctx_workspace_grew(self.h))

    def set_workspace_strict(self, on: bool):
        _check(self.lib.cat_ctx_set_workspace_strict(self.h, 1 if on else 0))

    # -- a1/a2 -----------------------------------------------------------------------------------------
    def ntt_forward(self, x: torch.Tensor, limb_ids=None, lazy=False):
        """x: [..., n_limbs, N], in place."""
        nl = x.shape[-2]
        npoly = x.numel() // (nl * self.n)
        _check(self.lib.cat_ntt_forward(self.h, self._p(x), self._ids(limb_ids), nl, npoly, int(lazy), self._stream()))
        return x

    def ntt_inverse(self, x: torch.Tensor, limb_ids=None, lazy=False):
        nl = x.shape[-2]
        npoly = x.numel() // (nl * self.n)
        _check(self.lib.cat_ntt_inverse(self.h, self._p(x), self._ids(limb_ids), nl, npoly, int(lazy), self._stream()))
        return x

    # -- a4/a5/a6 ---------------------------------------------------------------------------------------
    def _binary(self, fn, a, b, out, limb_ids):
        out = torch.empty_like(a) if out is None else out
        nl = a.shape[-2]
        npoly = a.numel() // (nl * self.n)
        _check(fn(self.h, self._p(a), self._p(b, a.shape), self._p(out, a.shape), self._ids(limb_ids), nl, npoly,
                  self._stream()))
        return out

    def _unary(self, fn, a, out, limb_ids):
        out = torch.empty_like(a) if out is None else out
        nl = a.shape[-2]
        npoly = a.numel() // (nl * self.n)
        _check(fn(self.h, self._p(a), self._p(out, a.shape), self._ids(limb_ids), nl, npoly, self._stream()))
        return out

    def _scalar(self, fn, a, scalars, out, limb_ids):
        out = torch.empty_like(a) if out is None else out
        nl = a.shape[-2]
        npoly = a.numel() // (nl * self.n)
        sc = (C.c_uint64 * nl)(*[int(s) for s in scalars])
        _check(fn(self.h, self._p(a), sc, self._p(out, a.shape), self._ids(limb_ids), nl, npoly, self._stream()))
        return out

    def add(self, a, b, out=None, limb_ids=None):
        return self._binary(self.lib.cat_add, a, b, out, limb_ids)

    def sub(self, a, b, out=None, limb_ids=None):
        return self._binary(self.lib.cat_sub, a, b, out, limb_ids)

    def dyadic(self, a, b, out=None, limb_ids=None):
        return self._binary(self.lib.cat_dyadic, a, b, out, limb_ids)

    def negate(self, a, out=None, limb_ids=None):
        return self._unary(self.lib.cat_negate, a, out, limb_ids)

    def modulo(self, a, out=None, limb_ids=None):
        return self._unary(self.lib.cat_modulo, a, out, limb_ids)

    def mul_scalar(self, a, scalars, out=None, limb_ids=None):
        return self._scalar(self.lib.cat_mul_scalar, a, scalars, out, limb_ids)

    def add_scalar(self, a, scalars, out=None, limb_ids=None):
        return self._scalar(self.lib.cat_add_scalar, a, scalars, out, limb_ids)

    def sub_scalar(self, a, scalars, out=None, limb_ids=None):
        return self._scalar(self.lib.cat_sub_scalar, a, scalars, out, limb_ids)

    # -- evaluator-level ----------------------------------------------------------------------------------
    def ckks_multiply(self, x, y, out=None):
        """x, y: [batch, 2, L, N] -> [batch, 3, L, N]"""
        b, _, L, _ = x.shape
        out = torch.empty((b, 3, L, self.n), dtype=torch.int64, device=x.device) if out is None else out
        _check(self.lib.cat_ckks_multiply(self.h, self._p(x, (b, 2, L, self.n)), self._p(y, (b, 2, L, self.n)),
                                        self._p(out, (b, 3, L, self.n)), L, b, self._stream()))
        return out

    def bgv_multiply(self, x, y, out=None):
        """x, y: [batch, 2, L, N] NTT form -> [batch, 3, L, N] (evaluator.cpp:710-841)"""
        b, _, L, _ = x.shape
        out = torch.empty((b, 3, L, self.n), dtype=torch.int64, device=x.device) if out is None else out
        _check(self.lib.cat_bgv_multiply(self.h, self._p(x, (b, 2, L, self.n)), self._p(y, (b, 2, L, self.n)),
                                        self._p(out, (b, 3, L, self.n)), L, b, self._stream()))
        return out

    def bgv_inv_q_last_mod_t(self, L: int) -> int:
        return int(self.lib.cat_bgv_inv_q_last_mod_t(self.h, L))

    def ckks_square(self, x, out=None):
        """x: [batch, 2, L, N] -> [batch, 3, L, N]"""
        b, _, L, _ = x.shape
        out = torch.empty((b, 3, L, self.n), dtype=torch.int64, device=x.device) if out is None else out
        _check(self.lib.cat_ckks_square(self.h, self._p(x, (b, 2, L, self.n)), self._p(out, (b, 3, L, self.n)), L, b,
                                        self._stream()))
        return out

    def _plain(self, fn, ct, plain, out):
        """ct: [batch, size, L, N]; plain: [L, N] (shared) or [batch, L, N]"""
        b, size, L, _ = ct.shape
        out = torch.empty_like(ct) if out is None else out
        stride = 0 if plain.dim() == 2 else L * self.n
        if plain.shape[-2] != L or not (plain.dim() == 2 or (plain.dim() == 3 and plain.shape[0] == b)):
            raise ValueError("plain is [L, N] (shared) or [batch, L, N]")
        _check(fn(self.h, self._p(ct), self._p(plain), stride, self._p(out, ct.shape), L, size, b, self._stream()))
        return out

    def multiply_plain_ntt(self, ct, plain, out=None):
        return self._plain(self.lib.cat_multiply_plain_ntt, ct, plain, out)

    def add_plain(self, ct, plain, out=None):
        return self._plain(self.lib.cat_add_plain, ct, plain, out)

    def sub_plain(self, ct, plain, out=None):
        return self._plain(self.lib.cat_sub_plain, ct, plain, out)

    def bfv_multiply(self, x, y, out=None):
        b, _, L, _ = x.shape
        out = torch.empty((b, 3, L, self.n), dtype=torch.int64, device=x.device) if out is None else out
        _check(self.lib.cat_bfv_multiply(self.h, self._p(x, (b, 2, L, self.n)), self._p(y, (b, 2, L, self.n)),
                                        self._p(out, (b, 3, L, self.n)), L, b, self._stream()))
        return out

    # -- BEHZ pieces (util/rns.cpp:903-1131), one level with L limbs ------------------------------------
    def bfv_base_Bsk(self, L: int):
        n = self.lib.cat_bfv_base_Bsk_size(self.h, L)
        out = (C.c_uint64 * max(n, 1))()
        _check(self.lib.cat_bfv_base_Bsk(self.h, L, out))
        return [int(out[i]) for i in range(n)]

    def _piece(self, fn, x, L, in_limbs, out_limbs):
        if x.shape[-2] != in_limbs:
            raise ValueError(f"expected {in_limbs} limbs, got {x.shape[-2]}")
        npoly = x.numel() // (in_limbs * self.n)
        out = torch.empty(tuple(x.shape[:-2]) + (out_limbs, self.n), dtype=torch.int64, device=x.device)
        _check(fn(self.h, self._p(x), self._p(out), L, npoly, self._stream()))
        return out

    def fastbconv_m_tilde(self, x, L):
        """[..., L, N] in base q -> [..., |Bsk| + 1, N] in Bsk U {m_tilde}"""
        bsk = self.lib.cat_bfv_base_Bsk_size(self.h, L)
        return self._piece(self.lib.cat_fastbconv_m_tilde, x, L, L, bsk + 1)

    def sm_mrq(self, x, L):
        bsk = self.lib.cat_bfv_base_Bsk_size(self.h, L)
        return self._piece(self.lib.cat_sm_mrq, x, L, bsk + 1, bsk)

    def fast_floor(self, x, L):
        bsk = self.lib.cat_bfv_base_Bsk_size(self.h, L)
        return self._piece(self.lib.cat_fast_floor, x, L, L + bsk, bsk)

    def fastbconv_sk(self, x, L):
        bsk = self.lib.cat_bfv_base_Bsk_size(self.h, L)
        return self._piece(self.lib.cat_fastbconv_sk, x, L, bsk, L)

    def switch_key(self, enc, target, key):
        """enc: [batch, 2, L, N] in place; target: [batch, L, N]; key: [k-1, 2, k, N]"""
        b, _, L, _ = enc.shape
        _check(self.lib.cat_switch_key(self.h, self._p(enc, (b, 2, L, self.n)), self._p(target, (b, L, self.n)),
                                       self._key(key), L, b, self._stream()))
        return enc

    def key_broadcast(self, key, comm=None, root=0, bucket_words=0):
        """cat_key_broadcast: replicate `key` from rank `root` of the RCCL communicator `comm` (an ncclComm_t as an
        integer / c_void_p; None = single process, no-op), in place."""
        _check(self.lib.cat_key_broadcast(self.h, C.c_void_p(comm) if isinstance(comm, int) else comm, self._p(key),
                                          key.numel(), root, bucket_words, self._stream()))
        return key

    def result_gather(self, local, comm=None, world=1, out=None):
        """cat_result_gather: every rank's `local` slice (equal sizes) in rank order -> [world, *local.shape]; comm = an
        ncclComm_t (integer / c_void_p) of `world` ranks, None = single process."""
        import torch

        if not local.is_contiguous():
            raise ValueError("local slice must be contiguous")
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        if out.numel() != world * local.numel() or not out.is_contiguous():
            raise ValueError("destination must hold world x local words")
        _check(self.lib.cat_result_gather(self.h, C.c_void_p(comm) if isinstance(comm, int) else comm, self._p(local),
                                          local.numel(), self._p(out), self._stream()))
        return out

    def relinearize(self, ct3, key):
        """ct3: [batch, 3, L, N]; returns a view of the first two polys (updated in place)."""
        b, _, L, _ = ct3.shape
        _check(self.lib.cat_relinearize(self.h, self._p(ct3, (b, 3, L, self.n)), self._key(key), L, b, self._stream()))
        return ct3[:, :2]

    def relinearize_rescale(self, ct3, key, out=None):
        """ct3: [batch, 3, L, N] (scratch afterwards) -> [batch, 2, L-1, N] = rescale(relinearize(ct3))"""
        b, _, L, _ = ct3.shape
        out = torch.empty((b, 2, L - 1, self.n), dtype=torch.int64, device=ct3.device) if out is None else out
        _check(self.lib.cat_relinearize_rescale(self.h, self._p(ct3, (b, 3, L, self.n)), self._key(key),
                                                self._p(out, (b, 2, L - 1, self.n)), L, b, self._stream()))
        return out

    def rescale(self, ct, out=None):
        b, size, L, _ = ct.shape
        out = torch.empty((b, size, L - 1, self.n), dtype=torch.int64, device=ct.device) if out is None else out
        _check(self.lib.cat_rescale(self.h, self._p(ct), self._p(out, (b, size, L - 1, self.n)), L, size, b, self._stream()))
        return out

    NOALLOC_SCRATCH_BYTES = 64 << 20  # CAT_NOALLOC_SCRATCH_BYTES (include/cat_amd.h)

    def rescale_noalloc(self, ct, out):
        """cat_rescale_noalloc: `out` is required (this path allocates nothing, torch included)."""
        b, size, L, _ = ct.shape
        _check(self.lib.cat_rescale_noalloc(self.h, self._p(ct), self._p(out, (b, size, L - 1, self.n)), L, size, b,
                                            self._stream()))
        return out

    def rescale_strided(self, ct, size, out):
        """ct: [batch, S, L, N] of which the first `size` polys are the ciphertext (stride S*L*N)."""
        b, S, L, _ = ct.shape
        if out.dim() != 4 or out.shape[0] != b or out.shape[1] < size or out.shape[2] != L - 1 or size > S:
            raise ValueError("out is [batch, >= size, L-1, N]")
        _check(self.lib.cat_rescale_strided(self.h, self._p(ct), S * L * self.n, self._p(out),
                                            out.shape[1] * (L - 1) * self.n, L, size, b, self._stream()))
        return out

    def mod_drop(self, ct, out=None):
        b, size, L, _ = ct.shape
        out = torch.empty((b, size, L - 1, self.n), dtype=torch.int64, device=ct.device) if out is None else out
        if out.numel() < b * size * (L - 1) * self.n:
            raise ValueError("out is too small")
        _check(self.lib.cat_mod_drop(self.h, self._p(ct), self._p(out), L, size, b, self._stream()))
        return out

    def apply_galois_ntt(self, x, elt, out=None):
        out = torch.empty_like(x) if out is None else out
        nl = x.shape[-2]
        npoly = x.numel() // (nl * self.n)
        _check(self.lib.cat_apply_galois_ntt(self.h, self._p(x), self._p(out, x.shape), elt, nl, npoly, self._stream()))
        return out

    def apply_galois(self, x, elt, out=None, limb_ids=None):
        out = torch.empty_like(x) if out is None else out
        nl = x.shape[-2]
        npoly = x.numel() // (nl * self.n)
        _check(self.lib.cat_apply_galois(self.h, self._p(x), self._p(out, x.shape), elt, self._ids(limb_ids), nl, npoly,
                                         self._stream()))
        return out

    def apply_galois_inplace(self, ct, elt, key):
        b, _, L, _ = ct.shape
        _check(self.lib.cat_apply_galois_inplace(self.h, self._p(ct, (b, 2, L, self.n)), elt, self._key(key), L, b,
                                                 self._stream()))
        return ct

    def apply_galois_rescale(self, ct, elt, key, out=None):
        """ct: [batch, 2, L, N] (scratch afterwards) -> [batch, 2, L-1, N] = rescale(apply_galois(ct))"""
        b, _, L, _ = ct.shape
        out = torch.empty((b, 2, L - 1, self.n), dtype=torch.int64, device=ct.device) if out is None else out
        _check(self.lib.cat_apply_galois_rescale(self.h, self._p(ct, (b, 2, L, self.n)), elt, self._key(key),
                                                 self._p(out, (b, 2, L - 1, self.n)), L, b, self._stream()))
        return out

    def hoisted_key_prepare(self, key, elt, out=None):
        """cat_hoisted_key_prepare: sigma_elt^-1 applied to every limb of a Galois key ([k-1, 2, k, N]), once per key"""
        out = torch.empty_like(key) if out is None else out
        _check(self.lib.cat_hoisted_key_prepare(self.h, self._key(key), elt, self._key(out), self._stream()))
        return out

    def rotate_hoisted(self, ct, elts, prepared_keys, out=None):
        """cat_rotate_hoisted: ct [batch, 2, L, N] -> [len(elts), batch, 2, L, N], one output batch per Galois element;
        prepared_keys[r] = hoisted_key_prepare(key of elts[r], elts[r]).  Not the reference's words (see cat_amd.h):
        each output decrypts to what apply_galois_inplace's does."""
        b, _, L, _ = ct.shape
        r = len(elts)
        if len(prepared_keys) != r:
            raise ValueError("one prepared key per Galois element")
        out = torch.empty((r, b, 2, L, self.n), dtype=torch.int64, device=ct.device) if out is None else out
        c_elts = (C.c_uint32 * max(r, 1))(*elts)
        c_keys = (C.c_void_p * max(r, 1))(*[self._key(k) for k in prepared_keys])
        self._p(out, (r, b, 2, L, self.n))
        c_outs = (C.c_void_p * max(r, 1))(*[out[i].data_ptr() for i in range(r)])
        _check(self.lib.cat_rotate_hoisted(self.h, self._p(ct, (b, 2, L, self.n)), c_elts, c_keys, r, c_outs, L, b,
                                           self._stream()))
        return out

    def rotate_vector(self, ct, step, key):
        return self.apply_galois_inplace(ct, galois_elt_from_step(self.log_n, step), key)
