#!/usr/bin/env python3
"""Writes tests/golden/blake2xb_vectors.json: outputs of the BLAKE2Xb build made from the reference's own C files
(oracle/_ref/libblake2_ref.so, oracle/Makefile target `ref`; needs /root/reference at build time) on seeded inputs,
stored as data (first 16 bytes + SHA-256 of each output), so that the pin of the PRNG's hash travels without that
library.  Run here, in the container that holds the reference:  python3 tests/golden/make_blake2xb_vectors.py"""
import ctypes as C
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "libblake2_ref.so")
OUT = os.path.join(ROOT, "tests", "golden", "blake2xb_vectors.json")


def cases():
    """(outlen, data, key) of every vector: seeded, the same in the generator and in the test"""
    rng = np.random.default_rng(2024)
    for outlen in (1, 63, 64, 65, 130, 4096, 4097):
        for inlen in (0, 8, 128, 129, 300):
            for keylen in (0, 1, 64):
                data = rng.integers(0, 256, inlen, dtype=np.uint8).tobytes()
                key = rng.integers(0, 256, keylen, dtype=np.uint8).tobytes()
                yield outlen, data, key
    # the PRNG's buffer refills: BLAKE2Xb(4096 bytes; in = counter (8 bytes LE), key = seed (64 bytes)), randomgen.cpp:201-211
    seed = np.array([11, 22, 33, 44, 55, 66, 77, 88], dtype=np.uint64).tobytes()
    for counter in range(4):
        yield 4096, np.uint64(counter).tobytes(), seed


def main():
    ref = C.CDLL(REF)
    ref.blake2xb.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    rows = []
    for outlen, data, key in cases():
        out = C.create_string_buffer(outlen)
        assert ref.blake2xb(out, outlen, data, len(data), key if key else None, len(key)) == 0
        rows.append({"outlen": outlen, "inlen": len(data), "keylen": len(key), "head": out.raw[:16].hex(),
                     "sha256": hashlib.sha256(out.raw).hexdigest()})
    json.dump({"_what": "BLAKE2Xb outputs of the reference's vendored BLAKE2 C files on the seeded inputs of cases()",
               "vectors": rows}, open(OUT, "w"), indent=0)
    print("wrote", OUT, len(rows), "vectors")


if __name__ == "__main__":
    main()
