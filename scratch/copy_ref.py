"""What a plain device copy reaches on this box at the sizes of the streaming launches (read + written bytes)."""
import torch
for mb in (60, 240, 956):
    n = mb * 1000 * 1000
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    ts = []
    for _ in range(10):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("torch copy of %4d MB: min %.1f us = %.2f TB/s read+write" % (mb, min(ts), 2 * n / min(ts) / 1e6))
