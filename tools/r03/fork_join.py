"""Experiment: BFV multiply's two independent transform chains (q limbs, Bsk limbs; N = 16384, batch 64) on one stream
against two streams (fork / join), same box."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import cat_amd  # noqa: E402

log_n, n = 14, 16384
q = bench.get_primes_desc(2 * n, 49, 8)
bsk = bench.get_primes_desc(2 * n, 61, 10)
ctx = cat_amd.Context(log_n, q + bsk, cat_amd.SCHEME_NONE, device=0)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
a = bench.uniform_limbs(torch, q, (256,), n, dev, gen)
b = bench.uniform_limbs(torch, bsk, (256,), n, dev, gen)
ida, idb = list(range(8)), list(range(8, 18))
s2 = torch.cuda.Stream()


def seq():
    ctx.ntt_forward(a, limb_ids=ida, lazy=True)
    ctx.ntt_forward(b, limb_ids=idb, lazy=True)
    ctx.ntt_inverse(a, limb_ids=ida, lazy=True)
    ctx.ntt_inverse(b, limb_ids=idb, lazy=True)


def fork():
    e0 = torch.cuda.Event()
    e0.record()
    s2.wait_event(e0)
    ctx.ntt_forward(a, limb_ids=ida, lazy=True)
    ctx.ntt_inverse(a, limb_ids=ida, lazy=True)
    with torch.cuda.stream(s2):
        ctx.ntt_forward(b, limb_ids=idb, lazy=True)
        ctx.ntt_inverse(b, limb_ids=idb, lazy=True)
        e1 = torch.cuda.Event()
        e1.record()
    torch.cuda.current_stream().wait_event(e1)


def timed(f, reps=40):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for r in range(3):
    print(f"rep {r}: one stream {timed(seq):7.1f} us | two streams {timed(fork):7.1f} us", flush=True)
