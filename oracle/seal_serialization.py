"""CPU restatement of the reference's wire format, uncompressed.  TEST INFRASTRUCTURE ONLY (see
seal_oracle.h): imported by tests/ as the checker of the facade's save()/load().

Followed, paths relative to /root/reference/native/src/seal/: SEALHeader and Serialization::Save/Load
(serialization.h:58-160, serialization.cpp:224-452), DynArray (dynarray.h:664-730), Ciphertext (ciphertext.cpp:188-400),
Plaintext (plaintext.cpp:207-290), SecretKey (secretkey.cuh:140-260: saved as a Plaintext), PublicKey (publickey.h:84-190:
saved as its Ciphertext), KSwitchKeys (kswitchkeys.cpp:42-160), UniformRandomGeneratorInfo (randomgen.cpp:99-158),
parms_id (encryptionparams.cpp:124-158).  The reference holds no golden byte streams for these (its serialization
tests round-trip, native/tests/seal/serialization.cpp:50-180, whose header fields are checked in
tests/test_serialization.py): byte-level parity with a real SEAL file is unpinned.
"""
from __future__ import annotations

import struct

import numpy as np

from . import seal_oracle as so

MAGIC, HEADER_SIZE, VERSION = 0xA15E, 0x10, (4, 0)
SEED_MARKER = 0xFFFFFFFFFFFFFFFF
PRNG_INFO_BYTES = 16 + 1 + 64


def header(size: int, compr_mode: int = 0, version=VERSION) -> bytes:
    """serialization.h:76-94"""
    return struct.pack("<HBBBBHQ", MAGIC, HEADER_SIZE, version[0], version[1], compr_mode, 0, size)


def parse_header(b: bytes, off: int = 0):
    magic, hs, vmaj, vmin, compr, reserved, size = struct.unpack_from("<HBBBBHQ", b, off)
    if (vmaj, vmin) != VERSION and not (vmaj == 3 and vmin >= 4):
        raise RuntimeError("incompatible version")
    if magic != MAGIC or hs != HEADER_SIZE:
        raise RuntimeError("loaded SEALHeader is invalid")
    if compr != 0:
        raise ValueError("unsupported compression mode")
    return (vmaj, vmin), size


def parms_id(scheme: int, n: int, primes, plain_modulus: int) -> bytes:
    """encryptionparams.cpp:124-158: BLAKE2b-256 over [scheme, N, q_i..., t] as 64-bit words"""
    words = np.array([scheme, n] + [int(q) for q in primes] + [plain_modulus], dtype=np.uint64)
    return so.blake2b(words.tobytes(), 32)


def dynarray(words: np.ndarray) -> bytes:
    """dynarray.h:664-690 inside its own header"""
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    body = struct.pack("<Q", w.size) + w.tobytes()
    return header(16 + len(body)) + body


def parse_dynarray(b: bytes, off: int, bound: int = 0):
    _, size = parse_header(b, off)
    (count,) = struct.unpack_from("<Q", b, off + 16)
    if bound and count > bound:
        raise RuntimeError("unexpected size")
    if size != 24 + 8 * count:
        raise RuntimeError("invalid data size")
    words = np.frombuffer(b, dtype=np.uint64, count=count, offset=off + 24).copy()
    return words, off + size


def prng_info(seed, prng_type: int = 1) -> bytes:
    body = struct.pack("<B", prng_type) + np.array(seed, dtype=np.uint64).tobytes()
    return header(16 + len(body)) + body


class Params:
    """Which level a parms_id names: level L uses the first L primes of the key-level chain (context.cu:607-765)."""

    def __init__(self, scheme: int, n: int, primes, plain_modulus: int = 0):
        self.scheme, self.n, self.primes, self.t = scheme, n, [int(q) for q in primes], plain_modulus
        self.k = len(self.primes)
        self.ids = {L: parms_id(scheme, n, self.primes[:L], plain_modulus) for L in range(1, self.k + 1)}

    def level_of(self, pid: bytes) -> int:
        for L, v in self.ids.items():
            if v == pid:
                return L
        return 0


def save_ciphertext(p: Params, ct: np.ndarray, ntt_form: bool, scale: float = 1.0, correction_factor: int = 1) -> bytes:
    """ciphertext.cpp:227-286; ct: [size][L][N].  A size-2 ciphertext whose second polynomial starts with the seed
    marker is written as its first polynomial plus the generator info that follows the marker."""
    size, L, n = ct.shape
    flat = np.ascontiguousarray(ct, dtype=np.uint64).reshape(-1)
    body = p.ids[L] + struct.pack("<BQQQdQ", 1 if ntt_form else 0, size, n, L, scale, correction_factor)
    if size == 2 and int(flat[L * n]) == SEED_MARKER:
        body += dynarray(flat[: L * n]) + flat[L * n + 1:].tobytes()[:PRNG_INFO_BYTES]
    else:
        body += dynarray(flat)
    return header(16 + len(body)) + body


def load_ciphertext(p: Params, b: bytes, off: int = 0, octx=None):
    """ciphertext.cpp:288-400 -> (dict, next offset).  Seeded ciphertexts are expanded with octx (an oracle Context)."""
    version, size = parse_header(b, off)
    pos = off + 16
    pid = b[pos:pos + 32]
    ntt, csize, n, L, scale = struct.unpack_from("<BQQQd", b, pos + 32)
    pos += 32 + 1 + 24 + 8
    corr = 1
    if version[0] == 4:
        (corr,) = struct.unpack_from("<Q", b, pos)
        pos += 8
    lvl = p.level_of(pid)
    if not lvl or n != p.n or L != lvl or csize < 2:
        raise RuntimeError("ciphertext data is invalid")
    total = csize * L * n
    words, pos = parse_dynarray(b, pos, total)
    if words.size == L * n and csize == 2:
        _, isz = parse_header(b, pos)
        ptype = b[pos + 16]
        seed = np.frombuffer(b, dtype=np.uint64, count=8, offset=pos + 17)
        pos += isz
        if ptype != 1:
            raise RuntimeError("unsupported prng_type")
        words = np.concatenate([words, octx.seal_expand_seed(L, seed).reshape(-1)])
    if words.size != total:
        raise RuntimeError("ciphertext data is invalid")
    if pos - off != size:
        raise RuntimeError("invalid data size")
    return {"data": words.reshape(csize, L, n), "ntt_form": bool(ntt), "scale": scale, "correction_factor": corr,
            "L": L}, pos


def save_plaintext(pid: bytes, words: np.ndarray, scale: float) -> bytes:
    """plaintext.cpp:207-233"""
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    body = pid + struct.pack("<Qd", w.size, scale) + dynarray(w)
    return header(16 + len(body)) + body


def load_plaintext(b: bytes, off: int = 0):
    _, size = parse_header(b, off)
    pid = b[off + 16:off + 48]
    count, scale = struct.unpack_from("<Qd", b, off + 48)
    words, pos = parse_dynarray(b, off + 64, count)
    if words.size != count or pos - off != size:
        raise RuntimeError("plaintext data is invalid")
    return {"parms_id": pid, "data": words, "scale": scale}, pos


def save_secret_key(p: Params, sk: np.ndarray) -> bytes:
    """secretkey.cuh:140-156: the Plaintext sk_ at the key level, scale 1.0"""
    return save_plaintext(p.ids[p.k], sk, 1.0)


def save_kswitch_keys(p: Params, keys: dict, entries: int) -> bytes:
    """kswitchkeys.cpp:42-83; keys: index -> [k-1][2][k][N]; entries = keys_.size() of the reference (the key count for
    RelinKeys, N for GaloisKeys, keygenerator.cpp:188)"""
    body = p.ids[p.k] + struct.pack("<Q", entries)
    for i in range(entries):
        key = keys.get(i)
        body += struct.pack("<Q", 0 if key is None else key.shape[0])
        if key is not None:
            for j in range(key.shape[0]):
                body += save_ciphertext(p, key[j], True)
    return header(16 + len(body)) + body


def load_kswitch_keys(p: Params, b: bytes, off: int = 0, octx=None):
    _, size = parse_header(b, off)
    pos = off + 16
    if b[pos:pos + 32] != p.ids[p.k]:
        raise RuntimeError("KSwitchKeys data is invalid")
    (dim1,) = struct.unpack_from("<Q", b, pos + 32)
    pos += 40
    keys = {}
    for i in range(dim1):
        (dim2,) = struct.unpack_from("<Q", b, pos)
        pos += 8
        pks = []
        for _ in range(dim2):
            ct, pos = load_ciphertext(p, b, pos, octx)
            if ct["L"] != p.k or not ct["ntt_form"] or ct["data"].shape[0] != 2:
                raise RuntimeError("PublicKey data is invalid")
            pks.append(ct["data"])
        if pks:
            keys[i] = np.stack(pks)
    if pos - off != size:
        raise RuntimeError("invalid data size")
    return {"keys": keys, "entries": dim1}, pos
