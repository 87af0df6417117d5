"""Host-only known answers of the facade's seal::Modulus / CoeffModulus / PlainModulus (tests/cpp/test_modulus.cpp):
the values the reference's own unit tests assert (native/tests/seal/modulus.cpp) and its default parameter tables."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_facade_modulus_classes_reproduce_the_reference_known_answers():
    exe = os.path.join(ROOT, "tests", "cpp", "modulus_test")
    assert os.path.exists(exe), "make modulus_test (or __graft_entry__.build())"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "modulus ok" in r.stdout, r.stdout + r.stderr
