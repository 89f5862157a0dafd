"""STFT / mel at the other specialised sizes (compile-time engine) on the headline batch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
from tools.bench_configs import steady


def timeit(fn, warm=0, reps=0):
    return steady(lambda i: fn(), ramp_s=0.5)


g = torch.Generator(device="cuda").manual_seed(1)
y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
out = {}
for n_fft, hop in ((512, 128), (1024, 256), (400, 160), (2048, 512), (1536, 384)):
    T = 1 + 220500 // hop
    F = n_fft // 2 + 1
    ms = timeit(lambda: ap.stft(y, n_fft=n_fft, hop_length=hop), warm=2, reps=5)
    out[f"stft_{n_fft}_{hop}"] = dict(ms=ms, frames_per_s=256 * T / ms * 1e3, alg_GBps=(4 * hop + 8 * F) * 256 * T / ms / 1e6)
    ms = timeit(lambda: ap.melspectrogram(y, sr=22050, n_fft=n_fft, hop_length=hop, n_mels=80), warm=2, reps=5)
    out[f"mel_{n_fft}_{hop}"] = dict(ms=ms, frames_per_s=256 * T / ms * 1e3, alg_GBps=(4 * hop + 4 * 80) * 256 * T / ms / 1e6)
print(json.dumps(out, indent=1))
