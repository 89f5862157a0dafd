"""Headline mel kernel time against batch size: fixed cost (launch, table set-up, tail) vs rate."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
from tools.bench_configs import timeit
g = torch.Generator(device="cuda").manual_seed(1)
out = {}
for B in (1, 4, 16, 32, 64, 128, 256, 512, 1024):
    y = torch.randn((B, 220500), device="cuda", generator=g) * 0.1
    ms = timeit(lambda: ap.melspectrogram(y, sr=22050, n_fft=2048, hop_length=512, n_mels=128))
    out[B] = dict(ms=ms, frames_per_s=B * 431 / ms * 1e3)
    del y
print(json.dumps(out, indent=1))
