"""Headline-size mel / stft with every pad mode (the edge / reflect kernels only differ at clip edges)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
from tools.bench_configs import timeit
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
out = {}
for pm in ("constant", "reflect", "edge"):
    out[f"mel_{pm}_ms"] = timeit(lambda: ap.melspectrogram(y, sr=22050, n_fft=2048, hop_length=512, n_mels=128, pad_mode=pm))
    out[f"stft_{pm}_ms"] = timeit(lambda: ap.stft(y, n_fft=2048, hop_length=512, pad_mode=pm))
print(json.dumps(out, indent=1))
