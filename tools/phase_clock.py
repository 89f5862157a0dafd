"""Where the cycles of the g16 ISTFT / STFT workgroups go (diagnostic build with s_memtime stamps at phase
boundaries in wave 0 of every workgroup; never the product library).
  build:  python tools/phase_clock.py build       (CPU: build/libap_phase.so, -DAP_PHASE_CLOCK)
  run:    python tools/phase_clock.py {istft|stft}   (GPU)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "build", "libap_phase.so")
if sys.argv[1] == "build":
    csrc = os.path.join(ROOT, "mlx-audio-primitives_amd", "csrc")
    srcs = [os.path.join(csrc, s) for s in ("audioprims.hip", "stft16.hip", "istft16.hip", "host_builders.cpp")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DAP_PHASE_CLOCK", "-o", LIB] + srcs, cwd=csrc)
    print(LIB)
    sys.exit(0)
os.environ["AP_LIB_PATH"] = LIB
import importlib
import numpy as np
import torch
import mlx_audio_primitives_amd as ap
from mlx_audio_primitives_amd import _extension as ext
op = sys.argv[1]
g = torch.Generator(device="cuda").manual_seed(1)
L = 220500
ys = [torch.randn((256, L), device="cuda", generator=g) * 0.1 for _ in range(3)]
stft_padded_rows = importlib.import_module("mlx_audio_primitives_amd.stft").stft_padded_rows
if op == "gl":                        # the STFT with the Griffin-Lim projection in its store phase (cfg3: 64 x 5 s)
    Sm = [torch.rand((64, 1025, 216), device="cuda", generator=g) for _ in range(3)]
    fn = lambda i: ap.griffinlim(Sm[i % 3], n_iter=4, hop_length=512, length=110250, random_state=0)
    names = ["window", "load issue", "forward", "split", "prefetch", "store:lds write", "store:bar", "store:read+global", "-", "-", "-", "-"]
    reader = ext.lib().ap_phase_read_stft16
elif op == "istft":
    Ss = [stft_padded_rows(y, n_fft=2048, hop_length=512) for y in ys]
    fn = lambda i: ap.istft(Ss[i % 3], hop_length=512, length=L)
    names = ["stage:read", "merge", "forward", "window", "gather", "setup", "bar(half)", "bar(gather)", "stage:wait+write", "stage:bar", "stage:issue", "-"]
    reader = ext.lib().ap_phase_read_istft16
else:
    outs = [torch.empty((256, 1025, 432, 2), device="cuda") for _ in range(3)]
    fn = lambda i: stft_padded_rows(ys[i % 3], n_fft=2048, hop_length=512, out=outs[i % 3])
    names = ["window", "load issue", "forward", "split", "prefetch", "store:lds write", "store:bar", "store:read+global", "-", "-", "-", "-"]
    reader = ext.lib().ap_phase_read_stft16
for i in range(300 if op != "gl" else 30):
    fn(i)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 12, np.uint64)
reader.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert reader(buf.ctypes.data, buf.size) == 0
a = buf.reshape(256, 8, 12).astype(np.float64)
a = a[a.sum((1, 2)) > 0]                      # (a capped grid leaves rows empty)
tot = a.sum(2)
print(f"workgroups {a.shape[0]}; wave total cycles: median {np.median(tot):.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
print("  %-20s" % "phase" + "".join(f"  wave{w:<4d}" for w in range(8)) + "   (median cycles over workgroups; % of wave 0's total)")
for k, n in enumerate(names):
    if n == "-":
        continue
    med = np.median(a[:, :, k], axis=0)
    print("  %-20s" % n + "".join(f"{m:10.0f}" for m in med) + f"   {100 * med[0] / np.median(tot[:, 0]):5.1f} %")
