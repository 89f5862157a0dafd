"""Measure every BASELINE.json config on ONE MI355X and print a JSON report
(copied into profiles/ by hand).  Not the driver's bench (that is bench.py); this is
the evidence table for DESIGN.md / profiles/README.md.

  cfg2  batch=256 x 10 s @16 kHz   melspectrogram n_fft=400 hop=160 n_mels=80 (Whisper)
  cfg3  batch=64 x 5 s @22.05 kHz  stft -> istft round trip, and griffinlim(32 iterations)
  cfg4  batch=1024 x 10 s @48 kHz  resample_poly 48k->16k, then mfcc(n_mfcc=13)
  cfg5  batch=4096 x 30 s @16 kHz mel pipeline over 8 GPUs: the per-GPU shard, 512 x 30 s
  head  batch=256 x 10 s @22.05kHz melspectrogram n_fft=2048 hop=512 n_mels=128
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlx_audio_primitives_amd as ap  # noqa: E402


def timeit(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]          # median of 10 after 3 warm-ups (benchmarks/utils.py:30-63)


def main():
    g = torch.Generator(device="cuda").manual_seed(42)
    rep = {}
    # headline
    y = torch.randn((256, 220500), device="cuda", generator=g) * 0.1
    ms = timeit(lambda: ap.melspectrogram(y, sr=22050, n_fft=2048, hop_length=512, n_mels=128))
    T = 431
    rep["headline_mel2048"] = dict(ms=ms, frames_per_s=256 * T / ms * 1e3,
                                   alg_GBps=(4 * 512 + 4 * 128) * 256 * T / ms / 1e6)
    ms = timeit(lambda: ap.stft(y, n_fft=2048, hop_length=512))
    rep["headline_stft2048"] = dict(ms=ms, frames_per_s=256 * T / ms * 1e3,
                                    alg_GBps=(4 * 512 + 8 * 1025) * 256 * T / ms / 1e6)
    del y
    # cfg2 whisper
    y = torch.randn((256, 160000), device="cuda", generator=g) * 0.1
    ms = timeit(lambda: ap.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=80))
    rep["cfg2_whisper_mel400"] = dict(ms=ms, frames_per_s=256 * 1001 / ms * 1e3,
                                      alg_GBps=(4 * 160 + 4 * 80) * 256 * 1001 / ms / 1e6)
    del y
    # cfg3 round trip + griffinlim
    y = torch.randn((64, 110250), device="cuda", generator=g) * 0.1
    S = ap.stft(y)
    ms_i = timeit(lambda: ap.istft(S, hop_length=512, length=110250))
    yr = ap.istft(S, hop_length=512, length=110250)
    rep["cfg3_istft2048"] = dict(ms=ms_i, frames_per_s=64 * 216 / ms_i * 1e3,
                                 alg_GBps=(8 * 1025 + 4 * 512) * 64 * 216 / ms_i / 1e6,
                                 round_trip_max_err=float((yr - y).abs().max()))
    mag = ap.magnitude(S)
    t0 = time.perf_counter()
    ms_g = timeit(lambda: ap.griffinlim(mag, n_iter=32, momentum=0.99, random_state=42, length=110250),
                  warm=1, reps=3)
    rep["cfg3_griffinlim32"] = dict(ms=ms_g, frame_iters_per_s=64 * 216 * 32 / ms_g * 1e3,
                                    alg_GBps=(36 * 1025 + 8 * 512) * 64 * 216 * 32 / ms_g / 1e6,
                                    note="includes host RNG + H2D of the initial phase, like the reference")
    del y, S, mag, yr
    # cfg4 resample + mfcc
    y = torch.randn((1024, 480000), device="cuda", generator=g) * 0.1
    ms_r = timeit(lambda: ap.resample_poly(y, 1, 3), warm=2, reps=5)
    y16 = ap.resample_poly(y, 1, 3)
    rep["cfg4_resample_poly_3to1"] = dict(ms=ms_r, out_samples_per_s=1024 * 160000 / ms_r * 1e3,
                                          alg_GBps=16 * 1024 * 160000 / ms_r / 1e6)
    del y
    ms_m = timeit(lambda: ap.mfcc(y16, sr=16000, n_mfcc=13, n_fft=2048, hop_length=512, n_mels=128),
                  warm=2, reps=5)
    rep["cfg4_mfcc13"] = dict(ms=ms_m, frames_per_s=1024 * 313 / ms_m * 1e3)
    del y16
    # cfg5: one GPU's shard of 4096 x 30 s @ 16 kHz (983 MB of samples)
    y = torch.randn((512, 480000), device="cuda", generator=g) * 0.1
    ms = timeit(lambda: ap.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=80), warm=2, reps=5)
    rep["cfg5_shard_whisper_mel400"] = dict(ms=ms, frames_per_s=512 * 3001 / ms * 1e3,
                                            alg_GBps=(4 * 160 + 4 * 80) * 512 * 3001 / ms / 1e6)
    ms = timeit(lambda: ap.melspectrogram(y, sr=16000, n_fft=2048, hop_length=512, n_mels=128), warm=2, reps=5)
    rep["cfg5_shard_mel2048"] = dict(ms=ms, frames_per_s=512 * 938 / ms * 1e3,
                                     alg_GBps=(4 * 512 + 4 * 128) * 512 * 938 / ms / 1e6)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
