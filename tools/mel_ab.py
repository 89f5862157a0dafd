"""A/B timing of the n_fft=2048 mel kernels on the headline batch: every variant in its own child
process (the switches are read once per process), interleaved rounds, median / min of the per-launch
HIP-event times.   usage: python tools/mel_ab.py [rounds]      (variants: env switches below)"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"wave(tile)": {"AP_MEL2048_WAVE": "1"}, "run": {}}
CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
import mlx_audio_primitives_amd as ap
from oracle import audio_oracle as ao
B, L = 256, 220500
g = torch.Generator(device="cuda").manual_seed(1)
t = torch.linspace(0, L / 22050, L, device="cuda")
chirp = torch.sin(2 * np.pi * (100 + 1000 * t) * t)
ys = [(chirp[None] + 0.1 * torch.randn((B, L), device="cuda", generator=g)).contiguous() for _ in range(3)]
kw = dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128)
rng = np.random.default_rng(0)
ysmall = rng.standard_normal((3, 30000)).astype(np.float32)
err = float(np.max(np.abs(ap.melspectrogram(torch.from_numpy(ysmall).cuda(), **kw).cpu().numpy() - ao.melspectrogram(ysmall, **kw))))
import time
t0 = time.time()
while time.time() - t0 < 1.0:                      # bring the clocks up (a cold GPU runs ~25 %% slower)
    for i in range(50): ap.melspectrogram(ys[i %% 3], **kw)
    torch.cuda.synchronize()
ts = []
for rep in range(5):                                # back-to-back streams of 200 launches
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): ap.melspectrogram(ys[i %% 3], **kw)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 200)
ts.sort()
print(json.dumps({"median_ms": ts[len(ts)//2], "min_ms": ts[0], "max_abs_err_vs_oracle": err}))
''' % ROOT

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {k: [] for k in VARIANTS}
for r in range(rounds):
    for name, env in VARIANTS.items():
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-1500:]); continue
        res[name].append(json.loads(line[-1]))
frames = 256 * 431
for name, rs in res.items():
    if rs:
        med = sorted(r["median_ms"] for r in rs)[len(rs) // 2]
        print(f"{name:12s} median {med:.4f} ms  min {min(r['min_ms'] for r in rs):.4f} ms  "
              f"{frames / med / 1e3:.1f} M frames/s  err {max(r['max_abs_err_vs_oracle'] for r in rs):.2e}")
