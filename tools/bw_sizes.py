"""Copy / fill / read bandwidth by buffer size, and the batched NTT by batch size (is the large-buffer rate a
paging effect?).  Usage: python3 tools/bw_sizes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cat_amd
from bench import get_primes_desc, uniform_limbs, LOG_N, N

def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for gib in (0.25, 1, 4, 16):
    w = int(gib * (1 << 27))
    x = torch.empty(w, dtype=torch.int64, device="cuda"); y = torch.empty(w, dtype=torch.int64, device="cuda")
    x.fill_(3)
    c = t(lambda: y.copy_(x)); f = t(lambda: x.fill_(7)); r = t(lambda: x.sum())
    print(f"{gib:5.2f} GiB: copy {2*w*8/c/1e6:6.0f} GB/s r+w   fill {w*8/f/1e6:6.0f} GB/s   sum {w*8/r/1e6:6.0f} GB/s", flush=True)
    del x, y
    torch.cuda.empty_cache()

primes = get_primes_desc(2 * N, 60, 32)
ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_NONE, device=0)
gen = torch.Generator(device="cuda:0"); gen.manual_seed(1)
for polys in (16, 64, 256, 1024):
    data = uniform_limbs(torch, primes, (polys,), N, torch.device("cuda:0"), gen)
    f = t(lambda: ctx.ntt_forward(data), 3)
    print(f"NTT {polys:5d} polys ({data.numel()*8/2**30:5.2f} GiB): fwd {f:.3f} ms = {16*data.numel()/f/1e6:.0f} GB/s algorithmic", flush=True)
    del data
    torch.cuda.empty_cache()
