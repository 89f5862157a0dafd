"""Clock the chip holds while the n_fft=2048 mel kernels run (diagnostic build with in-kernel
s_memtime / s_memrealtime stamps; never the product library).
  build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DAP_DIAG_STAMPS -o build/libap_diag.so csrc/audioprims.hip csrc/host_builders.cpp
  run:    python tools/diag_clock.py          (variant via AP_MEL2048_WAVE=1 / AP_MEL2048_RUN8=1)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AP_LIB_PATH"] = os.path.join(ROOT, "build", "libap_diag.so")
import subprocess, threading, time
import torch
import mlx_audio_primitives_amd as ap
from mlx_audio_primitives_amd import _extension as ext

B, L = 256, 220500
g = torch.Generator(device="cuda").manual_seed(1)
ys = [(0.3 * torch.randn((B, L), device="cuda", generator=g)).contiguous() for _ in range(3)]
kw = dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128)
watts = []
def _sample():
    while not done:
        try:
            out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            for ln in out.splitlines():
                if "Power (W)" in ln:
                    watts.append(float(ln.split(":")[-1]))
        except Exception:
            pass
done = False
th = threading.Thread(target=_sample); th.start()
t0 = time.time()
n_launch = 0
while time.time() - t0 < 4.0:            # long enough for the power management to settle
    for i in range(50):
        ap.melspectrogram(ys[i % 3], **kw)
    torch.cuda.synchronize()
    n_launch += 50
el = time.time() - t0
done = True; th.join()
print(f"{n_launch} launches in {el:.2f} s = {el / n_launch * 1e3:.4f} ms per launch (host-paced); board power samples (W): {watts}")
n = 4 * 256 * 16
buf = np.zeros(n, np.uint64)
ext.lib().ap_diag_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert ext.lib().ap_diag_read_stamps(buf.ctypes.data, n) == 0
cyc, ticks = buf[0::4].astype(np.float64), buf[1::4].astype(np.float64)
r0, r1 = buf[2::4].astype(np.float64), buf[3::4].astype(np.float64)
ok = ticks > 0
ghz = cyc[ok] / (ticks[ok] * 10.0)
frames_per_wave = B * 431 / ok.sum()
print(f"waves {ok.sum()}  frame loop: median {np.median(cyc[ok]):.0f} cycles = {np.median(ticks[ok]) / 100:.1f} us; "
      f"clock median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}); "
      f"{np.median(cyc[ok]) / frames_per_wave:.0f} cycles per frame per wave")

# the LAST launch: when did the waves enter / leave their frame loops (100 MHz ticks -> us)
a0, a1 = r0[ok] / 100.0, r1[ok] / 100.0
base = a0.min()
print(f"last launch: loops start {np.percentile(a0 - base, [0, 50, 99, 100]).round(1)} us after the first one; "
      f"loops end {np.percentile(a1 - base, [0, 1, 50, 99, 100]).round(1)} us; loop length {np.percentile(a1 - a0, [0, 50, 100]).round(1)} us")
# per wave slot of the workgroup (worker = workgroup * waves + wave): who is fast, who is slow?
nw = int(os.environ.get("DIAG_WAVES", "8"))
L = (a1 - a0)
if len(L) % nw == 0:
    per = L.reshape(-1, nw)
    print("loop length by wave slot (median us):", np.median(per, axis=0).round(1))
    print("per workgroup: fastest wave", np.median(per.min(axis=1)).round(1), "slowest", np.median(per.max(axis=1)).round(1),
          "mean", np.median(per.mean(axis=1)).round(1))
    wg_mean = per.mean(axis=1)
    print("workgroup-mean loop length percentiles [0,10,50,90,100]:", np.percentile(wg_mean, [0, 10, 50, 90, 100]).round(1))
