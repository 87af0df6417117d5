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

    # BFV / BGV with plaintexts: the facade's fresh ciphertexts equal the oracle's (C oracle for the
    # encryption of zero, oracle/seal_plain.py for the plaintext part), and the oracle decrypts the facade's switched
    # product to the plaintext the facade decrypted
    from oracle import seal_plain as sp

    t = 65537
    primes = oracle.coeff_modulus_create(n, [40, 40, 40, 41])
    seed = [301, 302, 303, 304, 305, 306, 307, 308]
    gamma = oracle.get_primes(2 * n, 61, 2)[1]
    m1 = [(i * i * 7 + 3 * i + 11) % t for i in range(n)]
    m2 = [(i * 5 + (i % 7) * 1000 + 65000) % t for i in range(n // 2)]
    for tag, scheme in (("bfv", oracle.SCHEME_BFV), ("bgv", oracle.SCHEME_BGV)):
        bfv = tag == "bfv"
        octx = oracle.Context(log_n, primes, scheme, t)
        p = ser.Params(scheme, n, primes, t)
        q = primes[:L]
        sk = octx.seal_secret_key(seed)
        pk = octx.seal_public_key(sk, seed)

        def fresh(m, use_pk):
            z = octx.seal_encrypt_zero(L, seed, pk=pk if use_pk else None, sk=sk)
            m = list(m) + [0] * (n - len(m))  # absent coefficients are zero
            add = sp.bfv_scale_plain(m, q, t) if bfv else sp.bgv_lift_plain(m, q, t)
            if not bfv:
                add = np.stack([oracle.NTTTables(log_n, q[i]).forward(add[i]) for i in range(L)])
            for i in range(L):
                z[0, i] = (z[0, i].astype(object) + add[i].astype(object)) % q[i]
            return z.astype(np.uint64)

        c1, _ = ser.load_ciphertext(p, (tmp_path / f"facade_{tag}_c1.bin").read_bytes())
        c2, _ = ser.load_ciphertext(p, (tmp_path / f"facade_{tag}_c2.bin").read_bytes())
        assert (c1["data"] == fresh(m1, False)).all() and (c2["data"] == fresh(m2, True)).all(), tag
        assert c1["ntt_form"] == (not bfv)
        prod, _ = ser.load_ciphertext(p, (tmp_path / f"facade_{tag}_prod.bin").read_bytes())
        assert prod["L"] == L - 1
        rk = octx.seal_relin_key(sk, seed)
        mul = octx.bfv_multiply if bfv else octx.ckks_multiply
        exp_ct = octx.rescale(octx.relinearize(mul(c1["data"], c2["data"]), rk))
        assert (prod["data"] == exp_ct).all(), tag
        ph = octx.seal_dot_product(sk, prod["data"], not bfv)
        if bfv:
            dec = sp.bfv_decrypt_scale_and_round(ph, q[:-1], t, gamma)
        else:
            ph = np.stack([oracle.NTTTables(log_n, q[i]).inverse(ph[i]) for i in range(L - 1)])
            assert prod["correction_factor"] == pow(q[-1], -1, t)
            dec = sp.bgv_decrypt_modt(ph, q[:-1], t, prod["correction_factor"])
        got = np.frombuffer((tmp_path / f"facade_{tag}_dec.bin").read_bytes(), dtype=np.uint64)
        assert got.tolist() == dec, tag


def test_cpp_facade_is_built_and_links_only_the_c_abi():
    """CPU check: the binary exists after build() and the facade header includes nothing from HIP/torch."""
    hdr = open(os.path.join(ROOT, "cat_amd", "host", "seal_hip.hpp")).read()
    assert "hip/" not in hdr and "torch" not in hdr and "oracle" not in hdr
    assert os.path.exists(EXE), "make facade_test"
