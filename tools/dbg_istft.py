import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mlx_audio_primitives_amd as ap
from oracle import audio_oracle as ao
rng = np.random.default_rng(256)
B, L = 16, 66150
y = rng.standard_normal((B, L)).astype(np.float32)
for hop in (256, 512, 1024):
    for center in (True, False):
        S = ao.stft(y, n_fft=2048, hop_length=hop, center=center)
        Sd = torch.from_numpy(S.astype(np.complex64)).cuda()
        T = S.shape[-1]
        for length in (None, L - 1000, L + 3000):
            got = ap.istft(Sd, hop_length=hop, center=center, length=length).cpu().numpy()
            want = ao.istft(S, hop_length=hop, n_fft=2048, center=center, length=length)
            end = (T - 1) * hop + 2048 - (1024 if center else 0)
            d = np.abs(got - want)
            bad = np.argwhere(~(d <= 2e-5))
            inner = bad[(bad[:, 1] >= 64) & (bad[:, 1] < end - 64)]
            print(hop, center, length, "T", T, "end", end, "finite", np.isfinite(got).all(), "nbad", len(bad),
                  "inner", len(inner), inner[:4].tolist(), "tail nonzero", int(np.count_nonzero(got[:, end:])))
