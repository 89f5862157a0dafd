"""magnitude / phase of the headline STFT (256 x 1025 x 431 bins): ms per call, median of 5 x 50 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
S = ap.stft(y)
for name, fn in (("magnitude", ap.magnitude), ("phase", ap.phase)):
    for _ in range(20):
        fn(S)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(S)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50)
    ts.sort()
    n = S.numel()
    print(f"{name}: {ts[2]:.4f} ms = {12 * n / ts[2] / 1e9:.2f} TB/s over 12 B/bin")
