"""Steady-state timings of shapes beside the BASELINE configs (same protocol as tools/bench_configs.py).
usage: python tools/bench_secondary.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlx_audio_primitives_amd as ap  # noqa: E402
from bench_configs import N_ROT, rec, steady  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(9)
    rep = {}
    B, L = 256, 220500
    ys = [torch.randn((B, L), device="cuda", generator=g) * 0.1 for _ in range(N_ROT)]
    for n_fft, hop in ((512, 128), (400, 160), (256, 64), (1024, 256)):
        T = 1 + L // hop
        F = n_fft // 2 + 1
        ms = steady(lambda i: ap.stft(ys[i % N_ROT], n_fft=n_fft, hop_length=hop))
        rep[f"stft_{n_fft}_{hop}"] = rec(ms, B * T, (4 * hop + 8 * F) * B * T, unit="frames")
        Ss = [ap.stft(y, n_fft=n_fft, hop_length=hop) for y in ys[:2]]
        ms = steady(lambda i: ap.istft(Ss[i % 2], hop_length=hop, length=L))
        rep[f"istft_{n_fft}_{hop}"] = rec(ms, B * T, (4 * hop + 8 * F) * B * T, unit="frames")
        del Ss
    ms = steady(lambda i: ap.stft(ys[i % N_ROT], n_fft=2048, hop_length=512, pad_mode="reflect"))
    rep["stft_2048_512_reflect"] = rec(ms, B * 431, (4 * 512 + 8 * 1025) * B * 431, unit="frames")
    ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=22050, n_fft=2048, hop_length=512, n_mels=128, pad_mode="reflect"))
    rep["mel_2048_512_128_reflect"] = rec(ms, B * 431, (4 * 512 + 4 * 128) * B * 431, unit="frames")
    ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=22050, n_fft=512, hop_length=128, n_mels=64, pad_mode="reflect"))
    rep["mel_512_128_64_reflect"] = rec(ms, B * 1723, (4 * 128 + 4 * 64) * B * 1723, unit="frames")
    ms = steady(lambda i: ap.mfcc(ys[i % N_ROT], sr=16000, n_mfcc=13, n_fft=400, hop_length=160, n_mels=80))
    rep["mfcc13_400_160_80"] = rec(ms, B * 1379, (4 * 160 + 4 * 13) * B * 1379, unit="frames")
    ms = steady(lambda i: ap.resample(ys[i % N_ROT], 22050, 16000, res_type="fft"), n_launch=10)
    rep["resample_fft_22050_to_16000"] = rec(ms, B * 160000, 8 * B * (220500 + 160000) // 2, unit="output samples")
    ms = steady(lambda i: ap.resample(ys[i % N_ROT], 22050, 16000, res_type="linear"))
    rep["resample_linear_22050_to_16000"] = rec(ms, B * 160000, 4 * B * (220500 + 160000), unit="output samples")
    ms = steady(lambda i: ap.resample_poly(ys[i % N_ROT], 160, 147), n_launch=10)
    rep["resample_poly_160_147"] = rec(ms, B * 240000, 4 * B * (220500 + 240000), unit="output samples")
    ms = steady(lambda i: ap.resample_poly(ys[i % N_ROT], 1, 3, padtype="line"))
    rep["resample_poly_1_3_padtype_line"] = rec(ms, B * 73500, 4 * B * (220500 + 73500), unit="output samples")
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
