"""cfg4 resample_poly 48 kHz -> 16 kHz on 1024 x 480 000: steady-state ms per launch (2 rotating inputs).
usage: python tools/time_resample.py   (AP_DECIM_NO_UNROLL=1 outside: the rolled loop)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
g = torch.Generator(device="cuda").manual_seed(1)
ys = [torch.randn((1024, 480000), device="cuda", generator=g) * 0.1 for _ in range(2)]
fn = lambda i: ap.resample_poly(ys[i % 2], 1, 3)
t0 = time.time(); i = 0
while time.time() - t0 < 1.0:
    for _ in range(20):
        fn(i); i += 1
    torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn(i); i += 1
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 50)
ts.sort()
print(f"resample_poly 3:1: {ts[2]:.4f} ms per launch = {16 * 1024 * 160000 / ts[2] / 1e9:.2f} TB/s algorithmic")
