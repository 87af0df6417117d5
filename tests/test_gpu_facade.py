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


def test_cpp_facade_is_built_and_links_only_the_c_abi():
    """CPU check: the binary exists after build() and the facade header includes nothing from HIP/torch."""
    hdr = open(os.path.join(ROOT, "cat_amd", "host", "seal_hip.hpp")).read()
    assert "hip/" not in hdr and "torch" not in hdr and "oracle" not in hdr
    assert os.path.exists(EXE), "make facade_test"
