"""SURVEY 8f row 3 on the CPU: the oracle's restatement of the reference's wire format (oracle/seal_serialization.py).
The reference's own serialization tests check header fields and round trips (native/tests/seal/serialization.cpp:50-180,
ciphertext.cpp / relinkeys.cpp / galoiskeys.cpp save-load round trips); the same is done here, plus the seeded form."""
import hashlib
import struct

import numpy as np
import pytest

SEED = [5, 6, 7, 8, 9, 10, 11, 12]
LOG_N, N = 8, 256


@pytest.fixture(scope="module")
def ser(oracle):
    from oracle import seal_serialization

    return seal_serialization


@pytest.fixture(scope="module")
def setup(oracle, ser):
    primes = oracle.coeff_modulus_create(N, [40, 30, 30, 40])
    octx = oracle.Context(LOG_N, primes, oracle.SCHEME_CKKS)
    p = ser.Params(oracle.SCHEME_CKKS, N, primes)
    sk = octx.seal_secret_key(SEED)
    return octx, p, sk, primes


def test_header_fields(ser):
    # native/tests/seal/serialization.cpp:50-113: 16 bytes; magic 0xA15E, header size 0x10, version, mode none, size
    h = ser.header(256)
    assert len(h) == 16
    magic, hs, vmaj, vmin, compr, reserved, size = struct.unpack("<HBBBBHQ", h)
    assert (magic, hs, vmaj, vmin, compr, reserved, size) == (0xA15E, 0x10, 4, 0, 0, 0, 256)
    assert ser.parse_header(h) == ((4, 0), 256)
    assert ser.parse_header(ser.header(256, version=(3, 4)))[0] == (3, 4)  # 3.4 and above load (serialization.h:144-161)
    with pytest.raises(RuntimeError):
        ser.parse_header(ser.header(256, version=(3, 3)))
    with pytest.raises(RuntimeError):
        ser.parse_header(ser.header(256, version=(5, 0)))
    with pytest.raises(RuntimeError):
        ser.parse_header(b"\x00\x00" + h[2:])
    with pytest.raises(ValueError):
        ser.parse_header(ser.header(256, compr_mode=1))  # zlib/zstd are off in this fork


def test_parms_id_is_blake2b_of_the_parameter_words(setup, ser, oracle):
    octx, p, sk, primes = setup
    words = np.array([oracle.SCHEME_CKKS, N] + primes[:3] + [0], dtype=np.uint64)
    assert p.ids[3] == hashlib.blake2b(words.tobytes(), digest_size=32).digest()
    assert len(set(p.ids.values())) == len(primes)
    assert p.level_of(p.ids[2]) == 2 and p.level_of(b"\0" * 32) == 0


def test_ciphertext_round_trip_and_seeded_form(setup, ser, oracle):
    octx, p, sk, primes = setup
    L = 3
    ct = octx.uniform(L, 3, 77)
    b = ser.save_ciphertext(p, ct, True, scale=2.0 ** 40, correction_factor=1)
    assert len(b) == 16 + 32 + 1 + 24 + 8 + 8 + 16 + 8 + ct.size * 8
    got, end = ser.load_ciphertext(p, b)
    assert end == len(b) and (got["data"] == ct).all() and got["ntt_form"] and got["scale"] == 2.0 ** 40 and got["L"] == L
    # seeded: half the data + 81 bytes of generator info; loading re-expands the second polynomial
    plain = octx.seal_encrypt_zero_symmetric(L, sk, SEED, True, save_seed=False)
    seeded = octx.seal_encrypt_zero_symmetric(L, sk, SEED, True, save_seed=True)
    assert int(seeded[1, 0, 0]) == ser.SEED_MARKER and (seeded[0] == plain[0]).all()
    bs = ser.save_ciphertext(p, seeded, True)
    assert len(bs) == len(ser.save_ciphertext(p, plain, True)) - L * N * 8 + ser.PRNG_INFO_BYTES
    got, end = ser.load_ciphertext(p, bs, 0, octx)
    assert end == len(bs) and (got["data"] == plain).all()
    # coefficient-form seeded encryption: the stored seed expands to the coefficient form of a
    octx_b = oracle.Context(LOG_N, primes, oracle.SCHEME_BFV, 65537)
    pb = ser.Params(oracle.SCHEME_BFV, N, primes, 65537)
    plain = octx_b.seal_encrypt_zero_symmetric(L, sk, SEED, False, save_seed=False)
    seeded = octx_b.seal_encrypt_zero_symmetric(L, sk, SEED, False, save_seed=True)
    got, _ = ser.load_ciphertext(pb, ser.save_ciphertext(pb, seeded, False), 0, octx_b)
    # both decrypt to small noise under sk (c1 differs: NTT-sampled vs coefficient-sampled a)
    for c in (plain, got["data"]):
        ph = octx_b.seal_dot_product(sk, c, False)
        centred = np.where(ph[0] > primes[0] // 2, ph[0].astype(np.int64) - primes[0], ph[0].astype(np.int64))
        assert np.abs(centred).max() <= 21
    # malformed input
    with pytest.raises(RuntimeError):
        ser.load_ciphertext(ser.Params(oracle.SCHEME_BFV, N, primes, 17), b)  # unknown parms_id
    with pytest.raises(RuntimeError):
        ser.load_ciphertext(p, b[:60] + struct.pack("<Q", 4) + b[68:])  # limb count does not match the level
    with pytest.raises(Exception):
        ser.load_ciphertext(p, b[:-8])


def test_keys_round_trip(setup, ser, oracle):
    octx, p, sk, primes = setup
    k = len(primes)
    b = ser.save_secret_key(p, sk)
    got, end = ser.load_plaintext(b)
    assert end == len(b) and got["parms_id"] == p.ids[k] and got["scale"] == 1.0 and (got["data"] == sk.reshape(-1)).all()
    rk = octx.seal_relin_key(sk, SEED)
    b = ser.save_kswitch_keys(p, {0: rk}, 1)
    got, end = ser.load_kswitch_keys(p, b)
    assert end == len(b) and got["entries"] == 1 and (got["keys"][0] == rk).all()
    elt = 3
    gk = octx.seal_galois_key(sk, elt, SEED)
    idx = (elt - 1) // 2
    b = ser.save_kswitch_keys(p, {idx: gk}, N)  # GaloisKeys: N entries, absent ones have dimension 0
    got, end = ser.load_kswitch_keys(p, b)
    assert end == len(b) and got["entries"] == N and list(got["keys"]) == [idx] and (got["keys"][idx] == gk).all()
    assert len(b) == 16 + 32 + 8 + 8 * N + (k - 1) * len(ser.save_ciphertext(p, gk[0], True))
