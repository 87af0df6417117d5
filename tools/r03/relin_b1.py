"""One N = 32768, 15 + 1 prime relinearize at batch B (default 1), repeated: for a kernel trace of the single-ciphertext
latency shape.  usage: relin_b1.py [batch] [reps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import cat_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
LOG_N, N = 15, 32768
primes = bench.bfv_default_32768()
k, L = len(primes), len(primes) - 1
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_CKKS, device=0)
ctx.reserve(ctx.workspace_need(B))
key = bench.uniform_limbs(torch, primes, (k - 1, 2), N, dev, gen)
ct3 = bench.uniform_limbs(torch, primes[:L], (B, 3), N, dev, gen)
for _ in range(10):
    ctx.relinearize(ct3, key)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    ctx.relinearize(ct3, key)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e6)
ts.sort()
print(f"relinearize batch {B}: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us per call", flush=True)
