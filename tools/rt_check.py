import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
g = torch.Generator(device="cuda").manual_seed(42)
y0 = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
for rep in range(6):
    y = torch.randn((64, 110250), device="cuda", generator=g) * 0.1
    S = ap.stft(y)
    for _ in range(5):
        yr = ap.istft(S, hop_length=512, length=110250)
    d = (yr - y).abs()
    print(rep, float(d.max()), float(d.mean()), float(yr.abs().max()), float((S.abs() ** 2).sum()))
