"""Run one operator a few times on the headline-sized batch (for rocprofv3 passes).
usage: python3 tools/run_op.py {stft|istft|stftd|istftd|mel|mfcc400|mel1024|mel512|stft512|whisper|gl|mfcc|resample|resfft|reslin} [reps] [ramp seconds]
(stftd / istftd: the dense layout the C entry points ap_stft_f32 / ap_istft_f32 serve; a ramp > 0 runs the operator
back to back for that long first, so that a --kernel-trace --stats average is a warm-clock number)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mlx_audio_primitives_amd as ap

op = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ramp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if op in ("stftd", "istftd"):
    ap.set_spectrum_layout("dense")
    op = op[:-1]
g = torch.Generator(device="cuda").manual_seed(1)
if op == "whisper":
    y = torch.randn((256, 160000), device="cuda", generator=g) * 0.1
    fn = lambda: ap.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=80)
elif op == "resfft":
    y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
    fn = lambda: ap.resample(y, 22050, 16000, res_type="fft")
elif op == "reslin":
    y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
    fn = lambda: ap.resample(y, 22050, 16000, res_type="linear")
elif op == "resample":
    y = torch.randn((1024, 480000), device="cuda", generator=g) * 0.1
    fn = lambda: ap.resample_poly(y, 1, 3)
elif op == "mfcc":
    y = torch.randn((1024, 160000), device="cuda", generator=g) * 0.1
    fn = lambda: ap.mfcc(y, sr=16000, n_mfcc=13, n_fft=2048, hop_length=512, n_mels=128)
elif op == "mfcc400":
    y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
    fn = lambda: ap.mfcc(y, sr=16000, n_mfcc=13, n_fft=400, hop_length=160, n_mels=80)
elif op == "mel1024":
    y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
    fn = lambda: ap.melspectrogram(y, sr=22050, n_fft=1024, hop_length=256, n_mels=80)
elif op == "mel512":
    y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
    fn = lambda: ap.melspectrogram(y, sr=22050, n_fft=512, hop_length=128, n_mels=64)
elif op == "stft512":
    y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
    fn = lambda: ap.stft(y, n_fft=512, hop_length=128)
elif op == "gl":
    y = torch.randn((64, 110250), device="cuda", generator=g) * 0.1
    S = ap.magnitude(ap.stft(y))
    fn = lambda: ap.griffinlim(S, n_iter=4, hop_length=512, length=110250)
else:
    y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
    if op == "stft":
        fn = lambda: ap.stft(y, n_fft=2048, hop_length=512)
    elif op == "istft":
        S = ap.stft(y, n_fft=2048, hop_length=512)
        fn = lambda: ap.istft(S, hop_length=512, length=220500)
    else:
        fn = lambda: ap.melspectrogram(y, sr=22050, n_fft=2048, hop_length=512, n_mels=128)
t0 = time.time()
while time.time() - t0 < ramp:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
