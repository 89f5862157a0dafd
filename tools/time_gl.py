"""Griffin-Lim, config 3 (64 clips x 5 s, n_fft 2048 hop 512, 32 iterations): ms per call, median of 5.
usage: python tools/time_gl.py      (AP_GL_PROJECT_PASS=1 outside: the three-kernel loop)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.randn((64, 110250), device="cuda", generator=g) * 0.1
S = ap.magnitude(ap.stft(y))
fn = lambda: ap.griffinlim(S, n_iter=32, hop_length=512, length=110250, random_state=0)
out = fn()
for _ in range(3):
    fn()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        fn()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 4)
ts.sort()
# spectral convergence of the result against the target magnitudes
R = ap.magnitude(ap.stft(out))
sc = float(torch.linalg.norm(R - S) / torch.linalg.norm(S))
print(f"griffinlim32: {ts[2]:.3f} ms (min {ts[0]:.3f}); spectral convergence {sc:.4f}; checksum {float(out.double().abs().sum()):.6f}")
