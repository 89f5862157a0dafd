import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
from tools.bench_configs import timeit
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
S = ap.stft(y, n_fft=1024, hop_length=256)
print("istft 1024/256 ms", timeit(lambda: ap.istft(S, hop_length=256, length=220500)))
mag = ap.magnitude(ap.stft(y[:64, :110250], n_fft=1024, hop_length=256))
print("griffinlim32 1024/256 64x5s ms", timeit(lambda: ap.griffinlim(mag, n_iter=32, hop_length=256, random_state=0, length=110250), warm=1, reps=3))
