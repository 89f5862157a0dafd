"""Achievable HBM bandwidth on this box with plain torch kernels (calibration for the rooflines)."""
import json, torch
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = {}
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
    b = torch.empty_like(a)
    ms = t(lambda: b.copy_(a)); out[f"copy_{mb}MB_GBps"] = 2 * n * 4 / ms / 1e6
    ms = t(lambda: b.fill_(1.0)); out[f"fill_{mb}MB_GBps"] = n * 4 / ms / 1e6
    ms = t(lambda: a.sum()); out[f"sum_{mb}MB_GBps"] = n * 4 / ms / 1e6
    del a, b
print(json.dumps(out, indent=1))
