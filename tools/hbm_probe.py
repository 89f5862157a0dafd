"""What HBM sustains on this box for plain streams (torch elementwise kernels): fill, copy, read-reduce.
usage: python tools/hbm_probe.py"""
import time
import torch
n = 1 << 28            # 1 GiB of float32
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)


def t(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t0 = time.time()
while time.time() - t0 < 1.0:
    y.copy_(x)
torch.cuda.synchronize()
ms = t(lambda: y.fill_(1.0)); print(f"fill   {4 * n / ms / 1e9:.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy   {8 * n / ms / 1e9:.2f} TB/s read + written")
ms = t(lambda: x.sum()); print(f"sum    {4 * n / ms / 1e9:.2f} TB/s read")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add    {8 * n / ms / 1e9:.2f} TB/s read + written")
