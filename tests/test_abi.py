"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/cat_amd.h declares (no compute calls without a GPU), and pure host helpers behave."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cat_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cat_[a-zA-Z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cat_amd import _lib

    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cat_amd.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared


def test_galois_elt_from_step_matches_reference_vectors():
    # util/galois.cpp:53-95, vectors of tests/util/galois.cpp:29-42
    import json

    from cat_amd import _lib

    ka = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))
    lib = _lib.load()
    for step, elt in ka["galois_elt_from_step"]:
        assert lib.cat_galois_elt_from_step(3, step) == elt
    assert lib.cat_galois_elt_from_step(3, 4) == 0  # "step count too large"


def test_ctx_create_rejects_bad_parameters_without_touching_a_gpu():
    import ctypes as C

    from cat_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    primes = (C.c_uint64 * 1)(17)
    assert lib.cat_ctx_create(0, primes, 1, 2, 0, 0, C.byref(h)) == 1  # poly_modulus_degree is invalid
    assert lib.cat_ctx_create(17, primes, 1, 2, 0, 0, C.byref(h)) == 1
    assert lib.cat_ctx_create(3, primes, 0, 2, 0, 0, C.byref(h)) == 1  # coeff_modulus is invalid
    bad = (C.c_uint64 * 1)(19)  # 19 != 1 mod 16
    assert lib.cat_ctx_create(3, bad, 1, 2, 0, 0, C.byref(h)) == 1
    assert b"invalid modulus" in lib.cat_last_error()
    notprime = (C.c_uint64 * 1)(33)
    assert lib.cat_ctx_create(3, notprime, 1, 2, 0, 0, C.byref(h)) == 1
    big = (C.c_uint64 * 1)(1 << 62)
    assert lib.cat_ctx_create(3, big, 1, 2, 0, 0, C.byref(h)) == 1
    assert h.value is None


def test_product_package_does_not_import_the_oracle():
    for top in ("cat_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".cuh", ".h", ".sh")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert "seal_oracle" not in src and "oracle/" not in src.replace("oracle/).", ""), f
