"""The context workspace (include/cat_amd.h): cat_ctx_workspace_need(batch) is the maximum of the launchers' own
formulas (csrc/context.hpp ws_words_*), so a context that reserved it never grows -- checked here in strict mode, where
a growth would fail the call, over every entry point that takes temporaries at every level (BFV multiply,
rotate and the BEHZ piece were not covered) -- and growth inside a call is visible (cat_ctx_workspace_grew) or refused
(cat_ctx_set_workspace_strict).  Results are checked elsewhere (test_gpu_parity.py); this file is about allocation."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cat():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import cat_amd

    return cat_amd


def _uniform(torch, primes, prefix, n, seed):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    out = torch.empty(tuple(prefix) + (len(primes), n), dtype=torch.int64, device="cuda")
    for j, q in enumerate(primes):
        out[..., j, :] = torch.randint(0, q, tuple(prefix) + (n,), dtype=torch.int64, device="cuda", generator=gen)
    return out


@pytest.mark.parametrize("scheme_name", ["CKKS", "BGV", "BFV"])
@pytest.mark.parametrize("log_n", [10, 12])
def test_reserved_need_covers_every_entry_point_at_every_level(cat, oracle, scheme_name, log_n):
    import torch

    n, k, batch = 1 << log_n, 5, 3
    primes = list(reversed(oracle.get_primes(2 * n, 50, k)))
    scheme = getattr(cat, "SCHEME_" + scheme_name)
    ctx = cat.Context(log_n, primes, scheme, plain_modulus=65537 if scheme_name != "CKKS" else 0)
    need = ctx.workspace_need(batch)
    assert need > 0
    ctx.reserve(need)
    held = ctx.workspace_bytes()
    assert held >= need and ctx.workspace_grew() == 0  # reserving does not count as growth inside a call
    ctx.set_workspace_strict(True)
    key = _uniform(torch, primes, (k - 1, 2), n, 7)
    elt = cat.galois_elt_from_step(log_n, 1)
    prepared = ctx.hoisted_key_prepare(key, elt)
    for L in range(k - 1, 0, -1):
        x = _uniform(torch, primes[:L], (batch, 2), n, 10 + L)
        y = _uniform(torch, primes[:L], (batch, 2), n, 20 + L)
        if scheme_name == "BFV":
            prod = ctx.bfv_multiply(x, y)
            bsk = len(ctx.bfv_base_Bsk(L))
            ctx.fastbconv_sk(_uniform(torch, [min(primes)] * bsk, (batch * 3,), n, 5), L)
        elif scheme_name == "BGV":
            prod = ctx.bgv_multiply(x, y)
        else:
            prod = ctx.ckks_multiply(x, y)
        ctx.relinearize(prod.clone(), key)
        ctx.switch_key(x.clone(), y[:, 0].contiguous(), key)
        ctx.apply_galois_inplace(x.clone(), elt, key)
        ctx.rotate_hoisted(x, [elt, elt], [prepared, prepared])
        if L >= 2:
            ctx.relinearize_rescale(prod.clone(), key)
            ctx.apply_galois_rescale(x.clone(), elt, key)
            ctx.rescale(prod)                      # size 3, out of place
            r = prod.clone()
            ctx.rescale(r, out=r.view(-1)[: batch * 3 * (L - 1) * n].view(batch, 3, L - 1, n))  # aliased: staged
            d = prod.clone()
            ctx.mod_drop(d, out=d.view(-1)[: batch * 3 * (L - 1) * n].view(batch, 3, L - 1, n))  # in place: staged
    # a rescale may also start at the key level (mod switching a freshly made key-level object)
    ctx.rescale(_uniform(torch, primes, (batch, 3), n, 3))
    torch.cuda.synchronize()
    assert ctx.workspace_bytes() == held and ctx.workspace_grew() == 0
    ctx.close()


def test_strict_mode_refuses_to_grow_and_growth_is_counted(cat, oracle):
    import torch

    log_n, n, k = 12, 4096, 4
    primes = list(reversed(oracle.get_primes(2 * n, 50, k)))
    ctx = cat.Context(log_n, primes, cat.SCHEME_CKKS)
    key = _uniform(torch, primes, (k - 1, 2), n, 7)
    ct3 = _uniform(torch, primes[: k - 1], (2, 3), n, 8)
    ref = ct3.clone()
    ctx.set_workspace_strict(True)
    with pytest.raises(RuntimeError, match="workspace would have to grow"):
        ctx.relinearize(ct3, key)
    torch.cuda.synchronize()
    assert torch.equal(ct3, ref)  # refused before anything was launched
    assert ctx.workspace_grew() == 0 and ctx.workspace_bytes() == 0
    ctx.set_workspace_strict(False)
    ctx.relinearize(ct3, key)  # grows on demand: visible afterwards
    torch.cuda.synchronize()
    assert ctx.workspace_grew() == 1 and ctx.workspace_bytes() > 0
    ctx.relinearize(ref.clone(), key)  # same size again: no further growth
    assert ctx.workspace_grew() == 1
    ctx.close()


def test_calls_leave_the_current_device_as_it_was(cat, oracle):
    """Every entry point makes the context's device current for its own duration only (cat_amd.h conventions).  A one-GPU
    box has one device, so this checks the no-switch branch and that context creation, calls and destruction keep it."""
    import torch

    primes = list(reversed(oracle.get_primes(2 * 1024, 40, 2)))
    before = torch.cuda.current_device()
    ctx = cat.Context(10, primes, cat.SCHEME_CKKS, device=0)
    x = _uniform(torch, primes[:1], (2,), 1024, 1)
    ctx.ntt_forward(x)
    ctx.ntt_inverse(x)
    ctx.close()
    assert torch.cuda.current_device() == before


def test_block_pool_hands_freed_blocks_out_again(cat, oracle):
    """cat_ctx_set_pool (gpu_memorypool.h:64-131 in the reference): with the pool on, cat_free keeps a block and the next
    cat_malloc of exactly that size gets it back; other sizes and a pool over its limit go to hipMalloc / hipFree."""
    import ctypes as C

    primes = list(reversed(oracle.get_primes(2 * 1024, 40, 2)))
    ctx = cat.Context(10, primes, cat.SCHEME_CKKS)
    lib, h = ctx.lib, ctx.h

    def malloc(nbytes):
        p = C.c_void_p()
        assert lib.cat_malloc(h, nbytes, C.byref(p)) == 0
        return p

    assert lib.cat_ctx_set_pool(h, 1 << 20) == 0
    a = malloc(65536)
    addr = a.value
    assert lib.cat_free(h, a) == 0
    b = malloc(65536)
    assert b.value == addr                     # the kept block
    c = malloc(65536)
    assert c.value != addr                     # nothing kept any more: a fresh block
    d = malloc(131072)
    assert lib.cat_free(h, b) == 0 and lib.cat_free(h, c) == 0 and lib.cat_free(h, d) == 0
    big = malloc(2 << 20)                      # larger than the pool's limit: really freed
    assert lib.cat_free(h, big) == 0
    assert lib.cat_ctx_set_pool(h, 0) == 0     # releases what is held; cat_free = hipFree again
    e = malloc(65536)
    assert lib.cat_free(h, e) == 0
    ctx.close()


def test_copy_is_device_to_device_and_refuses_overlap(cat, oracle):
    import ctypes as C

    import torch

    primes = list(reversed(oracle.get_primes(2 * 1024, 40, 2)))
    ctx = cat.Context(10, primes, cat.SCHEME_CKKS)
    x = torch.arange(4096, dtype=torch.int64, device="cuda")
    y = torch.zeros_like(x)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert ctx.lib.cat_copy(ctx.h, C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), x.numel() * 8, s) == 0
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    assert ctx.lib.cat_copy(ctx.h, C.c_void_p(x.data_ptr() + 64), C.c_void_p(x.data_ptr()), 1024, s) == 1
    assert b"alias" in ctx.lib.cat_last_error()
    assert ctx.lib.cat_copy(ctx.h, None, None, 0, s) == 0
    ctx.close()
