"""Metric op with the batch split over two contexts on two streams: do the memory-bound pieces of one half (tensor product,
inverse transforms, finish steps) hide under the VALU-bound key switch of the other?  Same box, interleaved repeats:
  one context, batch B            (what bench.py times)
  two contexts, batch B/2 each, two streams, launches interleaved per call
  two contexts, batch B each      (weak form: twice the work in flight)
Prints ops/s for each."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import cat_amd  # noqa: E402

LOG_N, N, L = 16, 65536, 24
k = L + 1
primes = list(reversed(bench.get_primes_desc(2 * N, 60, k)))
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
key = bench.uniform_limbs(torch, primes, (k - 1, 2), N, dev, gen)


class Half:
    def __init__(self, b):
        self.ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_CKKS, device=0)
        self.ctx.reserve(self.ctx.workspace_need(b))
        self.x = bench.uniform_limbs(torch, primes[:L], (b, 2), N, dev, gen)
        self.y = bench.uniform_limbs(torch, primes[:L], (b, 2), N, dev, gen)
        self.prod = torch.empty((b, 3, L, N), dtype=torch.int64, device=dev)
        self.out = torch.empty((b, 2, L - 1, N), dtype=torch.int64, device=dev)
        self.stream = torch.cuda.Stream()

    def mul(self):
        self.ctx.ckks_multiply(self.x, self.y, out=self.prod)

    def relin(self):
        self.ctx.relinearize_rescale(self.prod, key, out=self.out)


def timed(fn, ops_per_call, steps=20, settle=24):
    for _ in range(settle):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return ops_per_call * steps / (time.perf_counter() - t0)


one = Half(B)
a, b = Half(B // 2), Half(B // 2)
c, d = Half(B), Half(B)
torch.cuda.synchronize()


def single():
    one.mul()
    one.relin()


def pair_of(p, q, skew):
    def f():
        # skew: the second stream starts its key switch while the first is in its tensor product of the NEXT step
        with torch.cuda.stream(p.stream):
            p.mul()
        with torch.cuda.stream(q.stream):
            if not skew:
                q.mul()
            else:
                q.relin()
        with torch.cuda.stream(p.stream):
            p.relin()
        with torch.cuda.stream(q.stream):
            if not skew:
                q.relin()
            else:
                q.mul()
    return f


# same results whichever way: the half contexts against the whole one
one.x[: B // 2].copy_(a.x), one.y[: B // 2].copy_(a.y)
single()
torch.cuda.synchronize()
pair_of(a, b, False)()
c.mul(), d.mul()
torch.cuda.synchronize()
assert torch.equal(one.out[: B // 2], a.out), "split batch differs"
for rep in range(3):
    print(f"rep {rep}: one context batch {B}: {timed(single, B):8.1f} ops/s | two x {B // 2} on two streams: "
          f"{timed(pair_of(a, b, False), B):8.1f} | skewed: {timed(pair_of(a, b, True), B):8.1f} | two x {B} on two streams: "
          f"{timed(pair_of(c, d, False), 2 * B):8.1f} | skewed: {timed(pair_of(c, d, True), 2 * B):8.1f}", flush=True)
