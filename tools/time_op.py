"""Steady-state ms per launch of one operator (3 rotating inputs, 1 s ramp-up, median of 5 streams of
200 launches).  usage: python tools/time_op.py {whisper|mel|mel1024|mel512|stft512|stft|istft} [env switches outside]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
op = sys.argv[1]
if op in ("stftd", "istftd"):          # the dense-row layout the C entry points ap_stft_f32 / ap_istft_f32 serve
    ap.set_spectrum_layout("dense")
    op = op[:-1]
g = torch.Generator(device="cuda").manual_seed(1)
L = 160000 if op == "whisper" else 220500
ys = [torch.randn((256, L), device="cuda", generator=g) * 0.1 for _ in range(3)]
if op == "whisper":
    fn, units = (lambda i: ap.melspectrogram(ys[i % 3], sr=16000, n_fft=400, hop_length=160, n_mels=80)), 256 * 1001
elif op == "mel":
    fn, units = (lambda i: ap.melspectrogram(ys[i % 3], sr=22050, n_fft=2048, hop_length=512, n_mels=128)), 256 * 431
elif op == "mel1024":
    fn, units = (lambda i: ap.melspectrogram(ys[i % 3], sr=22050, n_fft=1024, hop_length=256, n_mels=80)), 256 * 862
elif op == "mel512":
    fn, units = (lambda i: ap.melspectrogram(ys[i % 3], sr=22050, n_fft=512, hop_length=128, n_mels=64)), 256 * 1723
elif op == "stft512":
    fn, units = (lambda i: ap.stft(ys[i % 3], n_fft=512, hop_length=128)), 256 * 1723
elif op == "stft1024":
    fn, units = (lambda i: ap.stft(ys[i % 3], n_fft=1024, hop_length=256)), 256 * 862
elif op == "stftrows":      # n_fft = 2048 STFT into rows padded to whole 128-byte lines (Griffin-Lim workspace layout)
    import importlib
    stft_padded_rows = importlib.import_module("mlx_audio_primitives_amd.stft").stft_padded_rows
    outs = [torch.empty((256, 1025, 432, 2), device="cuda") for _ in range(3)]
    fn, units = (lambda i: stft_padded_rows(ys[i % 3], n_fft=2048, hop_length=512, out=outs[i % 3])), 256 * 431
elif op == "stft":
    fn, units = (lambda i: ap.stft(ys[i % 3], n_fft=2048, hop_length=512)), 256 * 431
elif op == "istftrows":      # fused ISTFT from rows padded to whole 128-byte lines
    import importlib
    stft_padded_rows = importlib.import_module("mlx_audio_primitives_amd.stft").stft_padded_rows
    Ss = [stft_padded_rows(y, n_fft=2048, hop_length=512) for y in ys]
    fn, units = (lambda i: ap.istft(Ss[i % 3], hop_length=512, length=L)), 256 * 431
else:
    Ss = [ap.stft(y, n_fft=2048, hop_length=512) for y in ys]
    fn, units = (lambda i: ap.istft(Ss[i % 3], hop_length=512, length=L)), 256 * 431
t0 = time.time()
i = 0
while time.time() - t0 < 1.0:
    for _ in range(50):
        fn(i); i += 1
    torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn(i); i += 1
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 200)
ts.sort()
print(f"{op}: {ts[2]:.4f} ms per launch (min {ts[0]:.4f}) = {units / ts[2] * 1e3 / 1e9:.3f} G units/s")
