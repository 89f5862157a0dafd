"""Steady stream of headline-batch melspectrogram launches (for rocprofv3 --kernel-trace):
1 s of ramp-up, then N launches back to back.  usage: python3 tools/steady_mel.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator(device="cuda").manual_seed(1)
ys = [(0.3 * torch.randn((256, 220500), device="cuda", generator=g)).contiguous() for _ in range(3)]
kw = dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128)
t0 = time.time()
while time.time() - t0 < 1.0:
    for i in range(50): ap.melspectrogram(ys[i % 3], **kw)
    torch.cuda.synchronize()
for i in range(N): ap.melspectrogram(ys[i % 3], **kw)
torch.cuda.synchronize()
