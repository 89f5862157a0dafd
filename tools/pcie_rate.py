"""melspectrogram handed a HOST array (the Python API copies it in): the PCIe-inclusive rate DESIGN.md §5 notes.
usage: python tools/pcie_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mlx_audio_primitives_amd as ap
B, L = 256, 220500
y = np.random.default_rng(0).standard_normal((B, L)).astype(np.float32)
yp = torch.from_numpy(y).pin_memory()
for name, src in (("pageable numpy array", y), ("pinned torch tensor", yp)):
    for _ in range(3):
        ap.melspectrogram(src, sr=22050, n_fft=2048, hop_length=512, n_mels=128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        ap.melspectrogram(src, sr=22050, n_fft=2048, hop_length=512, n_mels=128)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name}: {ms:.2f} ms per call = {B * 431 / ms * 1e3 / 1e6:.1f} M frames/s ({B * L * 4 / ms / 1e6:.1f} GB/s over PCIe)")
