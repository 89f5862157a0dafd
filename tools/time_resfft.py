import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import mlx_audio_primitives_amd as ap
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
for _ in range(20): ap.resample(y, 22050, 16000, res_type="fft")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ap.resample(y, 22050, 16000, res_type="fft")
e1.record(); torch.cuda.synchronize()
print(os.environ.get("AP_CFFT_BUDGET"), e0.elapsed_time(e1) / 20, "ms")
