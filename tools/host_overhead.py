"""Host-side cost of one melspectrogram() call (tiny batch: GPU time negligible) and the
launch-to-launch time of the headline batch.  usage: python tools/host_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
kw = dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128)
y1 = torch.randn((1, 4096), device="cuda")
for _ in range(200): ap.melspectrogram(y1, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): ap.melspectrogram(y1, **kw)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host: {(t1 - t0) / 2000 * 1e6:.1f} us per call enqueued; drained after {(t2 - t1) * 1e3:.2f} ms more")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(500): ap.melspectrogram(y1, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
