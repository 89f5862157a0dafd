"""usage: python3 tools/run_istft.py n_fft hop [reps] : a few ISTFT launches on 256 x 10 s for rocprofv3 passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
n_fft, hop = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
y = torch.randn((256, 220500), device="cuda") * 0.1
S = ap.stft(y, n_fft=n_fft, hop_length=hop)
for _ in range(reps):
    ap.istft(S, hop_length=hop, length=220500)
torch.cuda.synchronize()
