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
# pure writes (what a kernel that mostly stores, like the Scharr planes at 4 B written per byte read, can hope for)
for mb in (60, 320, 956):
    n = mb * 1000 * 1000
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(10):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); a.fill_(7); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("torch fill of %4d MB: min %.1f us = %.2f TB/s written" % (mb, min(ts), n / min(ts) / 1e6))
# 1 B read per 4 B written, elementwise (the traffic shape of the Scharr kernel)
for mb in (80, 240):
    n = mb * 1000 * 1000
    a = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(10):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("torch u8 -> i32 convert of %4d M elements: min %.1f us = %.2f TB/s (5 B per element)" % (mb, min(ts), 5.0 * n / min(ts) / 1e6))
