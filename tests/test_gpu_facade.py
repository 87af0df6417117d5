"""Runs the C++ facade parity program (tests/cpp/test_facade.cpp): seal::Evaluator / seal::util names over
the C ABI, bit-exact against the oracle, plus the reference's exception behaviour."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "facade_test")


@pytest.mark.gpu
def test_cpp_facade_matches_oracle():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    assert os.path.exists(EXE), "build it first: make facade_test (or __graft_entry__.build())"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "facade ok" in r.stdout


@pytest.mark.gpu
def test_cpp_facade_wire_format_matches_oracle(tmp_path, oracle):
    """the facade writes the reference's wire format.  The oracle writes its files first (same
    parameters, same seed), the facade binary loads them, compares them with what it generated itself and writes its
    own; those must equal the oracle's byte for byte."""
    import numpy as np
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import seal_serialization as ser

    log_n, n, L = 12, 4096, 3
    primes = oracle.coeff_modulus_create(n, [50, 40, 40, 50])
    seed = [201, 202, 203, 204, 205, 206, 207, 208]
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    p = ser.Params(oracle.SCHEME_CKKS, n, primes)
    sk = octx.seal_secret_key(seed)
    elt = oracle.lib().so_galois_elt_from_step(log_n, 1)
    files = {
        "sk": ser.save_secret_key(p, sk),
        "pk": ser.save_ciphertext(p, octx.seal_public_key(sk, seed), True),
        "rk": ser.save_kswitch_keys(p, {0: octx.seal_relin_key(sk, seed)}, 1),
        "gk": ser.save_kswitch_keys(p, {(elt - 1) // 2: octx.seal_galois_key(sk, elt, seed)}, n),
        "ct": ser.save_ciphertext(p, octx.seal_encrypt_zero_symmetric(L, sk, seed, True), True, scale=2.0 ** 40),
        "seeded": ser.save_ciphertext(p, octx.seal_encrypt_zero_symmetric(L, sk, seed, True, save_seed=True), True),
        "plain": ser.save_plaintext(p.ids[L], octx.uniform(L, 1, 55)[0], 2.0 ** 40),
    }
    for name, data in files.items():
        (tmp_path / f"oracle_{name}.bin").write_bytes(data)
    r = subprocess.run([EXE, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "serialization: oracle files loaded" in r.stdout and "facade ok" in r.stdout
    for name, data in files.items():
        got = (tmp_path / f"facade_{name}.bin").read_bytes()
        assert got == data, f"{name}: {len(got)} bytes written, {len(data)} expected"


def test_cpp_facade_is_built_and_links_only_the_c_abi():
    """CPU check: the binary exists after build() and the facade header includes nothing from HIP/torch."""
    hdr = open(os.path.join(ROOT, "cat_amd", "host", "seal_hip.hpp")).read()
    assert "hip/" not in hdr and "torch" not in hdr and "oracle" not in hdr
    assert os.path.exists(EXE), "make facade_test"
