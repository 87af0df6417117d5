"""A short seeded slice of tests/fuzz_parity.py (randomised schemes, prime chains, levels, batch sizes), so the
round-end GPU run walks some shapes the fixed parity cases do not; the long run is `python3 tests/fuzz_parity.py`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [7001 + i for i in range(10)] + [7223, 7255, 7335, 9021, 9268])  # 72xx, 73xx: long prime chains; 9xxx: N = 2^16
def test_random_case_matches_oracle(oracle, seed):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import cat_amd

    from fuzz_parity import draw_case, run_case

    case = draw_case(np.random.default_rng(seed), 16 if seed >= 9000 else 13)
    assert run_case(cat_amd, oracle, case, seed) >= 13, case
