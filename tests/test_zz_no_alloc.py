"""cat_rescale_noalloc (include/cat_amd.h): the rescale entry point whose temporaries live in a static device array.
It has no counterpart in the reference; its contract is "the words of cat_rescale", so it is checked against that
entry point (itself parity-tested against the oracle) on two calls in a row, the second with a smaller input, and
once against the oracle directly."""
import numpy as np
import pytest
import torch

import cat_amd
from cat_amd.api import to_device, to_host

pytestmark = pytest.mark.gpu


def _uniform(primes, prefix, n, seed):
    rng = np.random.default_rng(seed)
    out = np.empty(tuple(prefix) + (len(primes), n), dtype=np.uint64)
    for j, q in enumerate(primes):
        out[..., j, :] = rng.integers(0, q, size=tuple(prefix) + (n,), dtype=np.uint64)
    return out


@pytest.mark.parametrize("log_n,bits", [(12, [50, 40, 40, 50]), (13, [60, 60, 60, 60, 60]), (10, [40, 30, 30, 40])])
def test_noalloc_rescale_equals_rescale_twice_in_a_row(oracle, log_n, bits):
    n = 1 << log_n
    primes = oracle.coeff_modulus_create(n, bits)
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_CKKS, device=0)
    octx = oracle.Context(log_n, primes, oracle.SCHEME_CKKS)
    L = len(primes) - 1
    # first call: batch 5 at the top data level; second call: a smaller input (batch 2, one level down)
    for call, (batch, lv) in enumerate([(5, L), (2, L - 1)]):
        x = _uniform(primes[:lv], (batch, 2), n, 100 + call)
        dx = to_device(x)
        want = ctx.rescale(dx)
        got = torch.full((batch, 2, lv - 1, n), -1, dtype=torch.int64, device="cuda:0")
        ctx.rescale_noalloc(dx, got)
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"call {call}: differs from cat_rescale"
        assert (to_host(got[0]) == octx.rescale(x[0])).all(), f"call {call}: differs from the oracle"
        assert torch.equal(dx, to_device(x)), "the input was modified"
    ctx.close()


def test_noalloc_rescale_size3_and_bfv(oracle):
    # size-3 ciphertexts (CKKS) and the BFV branch (coefficient form, no transform)
    n, log_n = 4096, 12
    primes = oracle.coeff_modulus_create(n, [50, 40, 40, 50])
    for scheme, t in ((cat_amd.SCHEME_CKKS, 0), (cat_amd.SCHEME_BFV, 65537), (cat_amd.SCHEME_BGV, 65537)):
        ctx = cat_amd.Context(log_n, primes, scheme, t, device=0)
        x = to_device(_uniform(primes[:3], (3, 3), n, 7))
        want = ctx.rescale(x)
        got = torch.empty_like(want)
        ctx.rescale_noalloc(x, got)
        assert torch.equal(got, want)
        ctx.close()


def test_noalloc_rescale_limits():
    n, log_n = 1 << 14, 14
    import bench

    primes = list(reversed(bench.get_primes_desc(2 * n, 60, 4)))
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_CKKS, device=0)
    L = 3
    per_ct = 2 * L * n * 8
    fits = cat_amd.Context.NOALLOC_SCRATCH_BYTES // per_ct
    x = torch.zeros((fits + 1, 2, L, n), dtype=torch.int64, device="cuda:0")
    out = torch.empty((fits + 1, 2, L - 1, n), dtype=torch.int64, device="cuda:0")
    ws_before = ctx.lib.cat_ctx_workspace_bytes(ctx.h)
    with pytest.raises(ValueError, match="exceed the static scratch"):
        ctx.rescale_noalloc(x, out)
    # the largest batch that fits runs, and equals cat_rescale
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(3)
    xs = torch.randint(0, min(primes), (fits, 2, L, n), dtype=torch.int64, device="cuda:0", generator=gen)
    got = torch.empty((fits, 2, L - 1, n), dtype=torch.int64, device="cuda:0")
    ctx.rescale_noalloc(xs, got)
    assert ctx.lib.cat_ctx_workspace_bytes(ctx.h) == ws_before, "the no-allocation path grew the context workspace"
    assert torch.equal(got, ctx.rescale(xs))
    # overlapping output is refused (cat_rescale's in-place form needs a staging copy this path does not have)
    with pytest.raises(ValueError, match="alias"):
        ctx.rescale_noalloc(xs, xs.view(-1)[: got.numel()].view(got.shape))
    ctx.close()
