"""Does the STFT write path care about row alignment?  n_fft = 512 / hop 128 with T = 1723 (rows at odd
multiples of 8 bytes) against T = 1728 (every row 64-byte aligned).  usage: python tools/stft_align_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
g = torch.Generator(device="cuda").manual_seed(1)
for n_fft, hop, L in ((512, 128, 220500), (512, 128, 221056), (512, 128, 220544), (400, 160, 160000), (400, 160, 163680)):
    ys = [torch.randn((256, L), device="cuda", generator=g) * 0.1 for _ in range(3)]
    fn = lambda i: ap.stft(ys[i % 3], n_fft=n_fft, hop_length=hop)
    T = fn(0).shape[-1]
    t0 = time.time(); i = 0
    while time.time() - t0 < 1.0:
        for _ in range(20):
            fn(i); i += 1
        torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn(i); i += 1
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 100)
    ts.sort()
    out_bytes = 256 * (n_fft // 2 + 1) * T * 8
    print(f"n_fft {n_fft} hop {hop} T {T} (T mod 8 = {T % 8}): {ts[2]:.4f} ms, output {out_bytes / 1e6:.0f} MB -> {out_bytes / ts[2] / 1e9:.2f} TB/s written")
