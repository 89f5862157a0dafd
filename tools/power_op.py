"""Board power while one operator runs back to back for 4 s.  usage: python tools/power_op.py {whisper|mel|...}"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap
op = sys.argv[1]
g = torch.Generator(device="cuda").manual_seed(1)
L = 160000 if op == "whisper" else 220500
ys = [torch.randn((256, L), device="cuda", generator=g) * 0.1 for _ in range(3)]
fns = {
    "whisper": lambda i: ap.melspectrogram(ys[i % 3], sr=16000, n_fft=400, hop_length=160, n_mels=80),
    "mel": lambda i: ap.melspectrogram(ys[i % 3], sr=22050, n_fft=2048, hop_length=512, n_mels=128),
    "mel1024": lambda i: ap.melspectrogram(ys[i % 3], sr=22050, n_fft=1024, hop_length=256, n_mels=80),
    "mel512": lambda i: ap.melspectrogram(ys[i % 3], sr=22050, n_fft=512, hop_length=128, n_mels=64),
    "stft": lambda i: ap.stft(ys[i % 3], n_fft=2048, hop_length=512),
}
fn = fns[op]
watts, done = [], False
def sample():
    while not done:
        out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"Power \(W\):\s*([\d.]+)", out)
        if m: watts.append(float(m.group(1)))
th = threading.Thread(target=sample); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for i in range(50): fn(n + i)
    torch.cuda.synchronize(); n += 50
el = time.time() - t0
done = True; th.join()
w = sorted(watts[len(watts) // 3:])
print(f"{op}: {el / n * 1e3:.4f} ms per launch; board power median {w[len(w) // 2]:.0f} W (samples {len(watts)})")
