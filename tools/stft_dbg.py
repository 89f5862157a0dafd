import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
from tools.bench_configs import timeit
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
out = {}
for rep in range(2):
    for dbg in ("0", "1", "2", "3", "7"):
        os.environ["AP_STFT_DBG"] = dbg
        out[f"dbg{dbg}_run{rep}"] = timeit(lambda: ap.stft(y, n_fft=2048, hop_length=512))
os.environ["AP_STFT_DBG"] = "0"
print(json.dumps(out, indent=1))
