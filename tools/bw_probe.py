import torch, time
x = torch.empty(1<<29, dtype=torch.int64, device='cuda')  # 4 GiB
y = torch.empty(1<<29, dtype=torch.int64, device='cuda')
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
ms=t(lambda: x.fill_(7)); print(f"fill 4GiB: {ms:.3f} ms  {x.numel()*8/ms/1e6:.0f} GB/s write")
ms=t(lambda: y.copy_(x)); print(f"copy 4GiB: {ms:.3f} ms  {2*x.numel()*8/ms/1e6:.0f} GB/s r+w")
ms=t(lambda: x.sum()); print(f"sum 4GiB: {ms:.3f} ms  {x.numel()*8/ms/1e6:.0f} GB/s read")
