"""Measure every BASELINE.json config on ONE MI355X and print a JSON report (the evidence table
of DESIGN.md / profiles/README.md; the driver's bench is bench.py).

  head  batch=256 x 10 s @22.05kHz melspectrogram n_fft=2048 hop=512 n_mels=128
  cfg2  batch=256 x 10 s @16 kHz   melspectrogram n_fft=400 hop=160 n_mels=80 (Whisper)
  cfg3  batch=64 x 5 s @22.05 kHz  stft -> istft round trip, and griffinlim(32 iterations)
  cfg4  batch=1024 x 10 s @48 kHz  resample_poly 48k->16k, then mfcc(n_mfcc=13)
  cfg5  batch=4096 x 30 s @16 kHz mel pipeline over 8 GPUs: the per-GPU shard, 512 x 30 s

Protocol: inputs resident in HBM; every operator rotates over N_ROT = 3 distinct
input buffers so that no launch re-reads a buffer that is still in the 256 MiB Infinity Cache
(cfg2's 164 MB input would otherwise sit inside it); 1 s of untimed ramp-up with the operator itself
(a fresh box idles at ~600 MHz); then 5 back-to-back streams of launches, HIP-event timed, median
stream / launches = ms per launch.  `frac_hbm` = algorithmic bytes / time / 8 TB/s.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlx_audio_primitives_amd as ap  # noqa: E402

N_ROT = 3
HBM = 8e12


def steady(fn, n_launch=None, ramp_s=1.0, streams=5):
    """fn(i) enqueues launch i.  Returns ms per launch in a back-to-back stream."""
    fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(1)
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-5)
    if n_launch is None:
        n_launch = max(3, min(200, int(0.03 / one)))          # ~30 ms per stream
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(n_launch):
            fn(i)
            i += 1
        torch.cuda.synchronize()
    ts = []
    for _ in range(streams):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_launch):
            fn(i)
            i += 1
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n_launch)
    ts.sort()
    return ts[len(ts) // 2]


def rec(ms, units, alg_bytes=None, **kw):
    d = dict(ms=ms, units_per_s=units / ms * 1e3)
    if alg_bytes is not None:
        d["alg_GBps"] = alg_bytes / ms / 1e6
        d["frac_hbm"] = alg_bytes / (ms * 1e-3) / HBM
    d.update(kw)
    return d


def main():
    g = torch.Generator(device="cuda").manual_seed(42)
    rep = {"protocol": "3 rotating inputs, 1 s ramp-up, median of 5 back-to-back streams (HIP events)"}

    def noise(B, L, n=N_ROT):
        return [torch.randn((B, L), device="cuda", generator=g) * 0.1 for _ in range(n)]

    # headline
    ys = noise(256, 220500)
    T = 431
    ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=22050, n_fft=2048, hop_length=512, n_mels=128))
    rep["headline_mel2048"] = rec(ms, 256 * T, (4 * 512 + 4 * 128) * 256 * T, unit="frames")
    ms = steady(lambda i: ap.stft(ys[i % N_ROT], n_fft=2048, hop_length=512))
    rep["headline_stft2048"] = rec(ms, 256 * T, (4 * 512 + 8 * 1025) * 256 * T, unit="frames")
    Ss = [ap.stft(y, n_fft=2048, hop_length=512) for y in ys]
    ms = steady(lambda i: ap.istft(Ss[i % N_ROT], hop_length=512, length=220500))
    rep["headline_istft2048"] = rec(ms, 256 * T, (8 * 1025 + 4 * 512) * 256 * T, unit="frames")
    del Ss
    ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=22050, n_fft=1024, hop_length=256, n_mels=80))
    T1 = 1 + 220500 // 256
    rep["mel1024_hop256_80"] = rec(ms, 256 * T1, (4 * 256 + 4 * 80) * 256 * T1, unit="frames")
    ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=22050, n_fft=512, hop_length=128, n_mels=64))
    T5 = 1 + 220500 // 128
    rep["mel512_hop128_64"] = rec(ms, 256 * T5, (4 * 128 + 4 * 64) * 256 * T5, unit="frames")
    ms = steady(lambda i: ap.stft(ys[i % N_ROT], n_fft=512, hop_length=128))
    rep["stft512_hop128"] = rec(ms, 256 * T5, (4 * 128 + 8 * 257) * 256 * T5, unit="frames")
    del ys
    # cfg2 whisper
    ys = noise(256, 160000)
    ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=16000, n_fft=400, hop_length=160, n_mels=80))
    rep["cfg2_whisper_mel400"] = rec(ms, 256 * 1001, (4 * 160 + 4 * 80) * 256 * 1001, unit="frames")
    del ys
    # cfg3 round trip + griffinlim
    ys = noise(64, 110250)
    Ss = [ap.stft(y) for y in ys]
    ms_s = steady(lambda i: ap.stft(ys[i % N_ROT]))
    rep["cfg3_stft2048"] = rec(ms_s, 64 * 216, (4 * 512 + 8 * 1025) * 64 * 216, unit="frames")
    ms_i = steady(lambda i: ap.istft(Ss[i % N_ROT], hop_length=512, length=110250))
    yr = ap.istft(Ss[0], hop_length=512, length=110250)
    rep["cfg3_istft2048"] = rec(ms_i, 64 * 216, (8 * 1025 + 4 * 512) * 64 * 216, unit="frames",
                                round_trip_max_err=float((yr - ys[0]).abs().max()))
    mags = [ap.magnitude(S) for S in Ss]
    ms_g = steady(lambda i: ap.griffinlim(mags[i % N_ROT], n_iter=32, momentum=0.99, random_state=42, length=110250),
                  n_launch=4, ramp_s=0.5, streams=3)
    rep["cfg3_griffinlim32"] = rec(ms_g, 64 * 216 * 32, (36 * 1025 + 8 * 512) * 64 * 216 * 32, unit="frame-iterations")
    del ys, Ss, mags, yr
    # cfg4 resample + mfcc
    ys = noise(1024, 480000, n=2)
    ms_r = steady(lambda i: ap.resample_poly(ys[i % 2], 1, 3), ramp_s=0.5)
    rep["cfg4_resample_poly_3to1"] = rec(ms_r, 1024 * 160000, 16 * 1024 * 160000, unit="output samples")
    y16 = [ap.resample_poly(y, 1, 3) for y in ys]
    del ys
    ms_m = steady(lambda i: ap.mfcc(y16[i % 2], sr=16000, n_mfcc=13, n_fft=2048, hop_length=512, n_mels=128), ramp_s=0.5)
    rep["cfg4_mfcc13"] = rec(ms_m, 1024 * 313, (4 * 512 + 4 * 13) * 1024 * 313, unit="frames")
    del y16
    # cfg5: one GPU's shard of 4096 x 30 s @ 16 kHz (983 MB of samples per buffer)
    ys = noise(512, 480000, n=2)
    ms = steady(lambda i: ap.melspectrogram(ys[i % 2], sr=16000, n_fft=400, hop_length=160, n_mels=80), ramp_s=0.5)
    rep["cfg5_shard_whisper_mel400"] = rec(ms, 512 * 3001, (4 * 160 + 4 * 80) * 512 * 3001, unit="frames")
    ms = steady(lambda i: ap.melspectrogram(ys[i % 2], sr=16000, n_fft=2048, hop_length=512, n_mels=128), ramp_s=0.5)
    rep["cfg5_shard_mel2048"] = rec(ms, 512 * 938, (4 * 512 + 4 * 128) * 512 * 938, unit="frames")
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
