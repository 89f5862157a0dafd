"""Every BASELINE.json config on ONE MI355X, steady state, plus the reference's published single-clip
rows.  Used two ways: `python tools/bench_configs.py` prints the report, and bench.py embeds the same
report as the `configs` object of its JSON line (after its own timed region, so `value` is untouched).

  head  batch=256 x 10 s @22.05kHz melspectrogram / stft / istft n_fft=2048 hop=512 n_mels=128
  cfg2  batch=256 x 10 s @16 kHz   melspectrogram n_fft=400 hop=160 n_mels=80 (Whisper)
  cfg3  batch=64 x 5 s @22.05 kHz  stft -> istft round trip, and griffinlim(32 iterations)
  cfg4  batch=1024 x 10 s @48 kHz  resample_poly 48k->16k, then mfcc(n_mfcc=13); and the chain as one call
  cfg5  batch=4096 x 30 s @16 kHz mel pipeline over 8 GPUs: the per-GPU shard, 512 x 30 s

Protocol: inputs resident in HBM; every operator rotates over N_ROT = 3 distinct
input buffers so that no launch re-reads a buffer that is still in the 256 MiB Infinity Cache
(cfg2's 164 MB input would otherwise sit inside it); `ramp_s` of untimed ramp-up with the operator itself
(a fresh box idles at ~600 MHz); then 5 back-to-back streams of launches, HIP-event timed, median
stream / launches = ms per launch.  `frac_hbm` = algorithmic bytes (SURVEY.md §8d's per-unit figures)
/ time / 8 TB/s.

Published rows (benchmarks/README.md:34-47,109-138 of the reference: 1 clip x 22 050 samples, median of
10 synchronised calls after 3 warm-ups, benchmarks/utils.py:30-63): `latency_ms` is the wall time of one
call including the host side and the synchronisation, the number a user of the reference's benchmark
table would compare; accuracy beside it as benchmarks/utils.py:66-89 reports it (max / mean abs error,
Pearson r) against the CPU oracle.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
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


def run_configs(ramp_s=0.5, which=None):
    """Steady-state report of the BASELINE configs.  `which`: iterable of section names
    ("headline", "cfg2", "cfg3", "cfg4", "cfg5", "other") or None for all."""
    which = set(which or ("headline", "cfg2", "cfg3", "cfg4", "cfg5", "other"))
    g = torch.Generator(device="cuda").manual_seed(42)
    rep = {"protocol": f"{N_ROT} rotating inputs, {ramp_s} s ramp-up per operator, median of 5 back-to-back "
                       "streams (HIP events on the launch stream); frac_hbm = SURVEY 8(d) algorithmic bytes / time / 8 TB/s"}

    def noise(B, L, n=N_ROT):
        return [torch.randn((B, L), device="cuda", generator=g) * 0.1 for _ in range(n)]

    T = 431
    if which & {"headline", "other"}:
        ys = noise(256, 220500)
    if "headline" in which:
        ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=22050, n_fft=2048, hop_length=512, n_mels=128), ramp_s=ramp_s)
        rep["headline_mel2048"] = rec(ms, 256 * T, (4 * 512 + 4 * 128) * 256 * T, unit="frames")
        ms = steady(lambda i: ap.stft(ys[i % N_ROT], n_fft=2048, hop_length=512), ramp_s=ramp_s)
        rep["headline_stft2048"] = rec(ms, 256 * T, (4 * 512 + 8 * 1025) * 256 * T, unit="frames")
        Ss = [ap.stft(y, n_fft=2048, hop_length=512) for y in ys]
        ms = steady(lambda i: ap.istft(Ss[i % N_ROT], hop_length=512, length=220500), ramp_s=ramp_s)
        rep["headline_istft2048"] = rec(ms, 256 * T, (8 * 1025 + 4 * 512) * 256 * T, unit="frames")
        del Ss
        # the same two operators on DENSE rows: what the C entry points ap_stft_f32 / ap_istft_f32 (the drop-in for the
        # reference's binding) read and write; `stft()` itself returns line-padded rows
        prev = ap.set_spectrum_layout("dense")
        try:
            ms = steady(lambda i: ap.stft(ys[i % N_ROT], n_fft=2048, hop_length=512), ramp_s=ramp_s)
            rep["headline_stft2048_dense_rows"] = rec(ms, 256 * T, (4 * 512 + 8 * 1025) * 256 * T, unit="frames")
            Ss = [ap.stft(y, n_fft=2048, hop_length=512) for y in ys]
            ms = steady(lambda i: ap.istft(Ss[i % N_ROT], hop_length=512, length=220500), ramp_s=ramp_s)
            rep["headline_istft2048_dense_rows"] = rec(ms, 256 * T, (8 * 1025 + 4 * 512) * 256 * T, unit="frames")
            del Ss
        finally:
            ap.set_spectrum_layout(prev)
    if "other" in which:
        ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=22050, n_fft=1024, hop_length=256, n_mels=80), ramp_s=ramp_s)
        T1 = 1 + 220500 // 256
        rep["mel1024_hop256_80"] = rec(ms, 256 * T1, (4 * 256 + 4 * 80) * 256 * T1, unit="frames")
        ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=22050, n_fft=512, hop_length=128, n_mels=64), ramp_s=ramp_s)
        T5 = 1 + 220500 // 128
        rep["mel512_hop128_64"] = rec(ms, 256 * T5, (4 * 128 + 4 * 64) * 256 * T5, unit="frames")
        ms = steady(lambda i: ap.stft(ys[i % N_ROT], n_fft=512, hop_length=128), ramp_s=ramp_s)
        rep["stft512_hop128"] = rec(ms, 256 * T5, (4 * 128 + 8 * 257) * 256 * T5, unit="frames")
    if which & {"headline", "other"}:
        del ys
    if "cfg2" in which:
        ys = noise(256, 160000)
        ms = steady(lambda i: ap.melspectrogram(ys[i % N_ROT], sr=16000, n_fft=400, hop_length=160, n_mels=80), ramp_s=ramp_s)
        rep["cfg2_whisper_mel400"] = rec(ms, 256 * 1001, (4 * 160 + 4 * 80) * 256 * 1001, unit="frames")
        del ys
    if "cfg3" in which:
        ys = noise(64, 110250)
        Ss = [ap.stft(y) for y in ys]
        ms_s = steady(lambda i: ap.stft(ys[i % N_ROT]), ramp_s=ramp_s)
        rep["cfg3_stft2048"] = rec(ms_s, 64 * 216, (4 * 512 + 8 * 1025) * 64 * 216, unit="frames")
        ms_i = steady(lambda i: ap.istft(Ss[i % N_ROT], hop_length=512, length=110250), ramp_s=ramp_s)
        yr = ap.istft(Ss[0], hop_length=512, length=110250)
        rep["cfg3_istft2048"] = rec(ms_i, 64 * 216, (8 * 1025 + 4 * 512) * 64 * 216, unit="frames",
                                    round_trip_max_err=float((yr - ys[0]).abs().max()))
        mags = [ap.magnitude(S) for S in Ss]
        ms_g = steady(lambda i: ap.griffinlim(mags[i % N_ROT], n_iter=32, momentum=0.99, random_state=42, length=110250),
                      n_launch=4, ramp_s=ramp_s, streams=3)
        rep["cfg3_griffinlim32"] = rec(ms_g, 64 * 216 * 32, (36 * 1025 + 8 * 512) * 64 * 216 * 32, unit="frame-iterations")
        del ys, Ss, mags, yr
    if "cfg4" in which:
        ys = noise(1024, 480000, n=2)
        ms_r = steady(lambda i: ap.resample_poly(ys[i % 2], 1, 3), ramp_s=ramp_s)
        rep["cfg4_resample_poly_3to1"] = rec(ms_r, 1024 * 160000, 16 * 1024 * 160000, unit="output samples")
        y16 = [ap.resample_poly(y, 1, 3) for y in ys]
        ms_m = steady(lambda i: ap.mfcc(y16[i % 2], sr=16000, n_mfcc=13, n_fft=2048, hop_length=512, n_mels=128), ramp_s=ramp_s)
        rep["cfg4_mfcc13"] = rec(ms_m, 1024 * 313, (4 * 512 + 4 * 13) * 1024 * 313, unit="frames")
        del y16
        # the whole config as the user writes it: 48 kHz in, 13 coefficients out; algorithmic bytes = every
        # 48 kHz sample read once (4 * 3 hop per frame) + the coefficients written (4 * 13)
        ms_c = steady(lambda i: ap.mfcc(ap.resample_poly(ys[i % 2], 1, 3), sr=16000, n_mfcc=13, n_fft=2048,
                                        hop_length=512, n_mels=128), ramp_s=ramp_s)
        rep["cfg4_chain_48k_to_mfcc13"] = rec(ms_c, 1024 * 313, (4 * 3 * 512 + 4 * 13) * 1024 * 313, unit="frames")
        del ys
    if "cfg5" in which:
        # one GPU's shard of 4096 x 30 s @ 16 kHz (983 MB of samples per buffer)
        ys = noise(512, 480000, n=2)
        ms = steady(lambda i: ap.melspectrogram(ys[i % 2], sr=16000, n_fft=400, hop_length=160, n_mels=80), ramp_s=ramp_s)
        rep["cfg5_shard_whisper_mel400"] = rec(ms, 512 * 3001, (4 * 160 + 4 * 80) * 512 * 3001, unit="frames")
        ms = steady(lambda i: ap.melspectrogram(ys[i % 2], sr=16000, n_fft=2048, hop_length=512, n_mels=128), ramp_s=ramp_s)
        rep["cfg5_shard_mel2048"] = rec(ms, 512 * 938, (4 * 512 + 4 * 128) * 512 * 938, unit="frames")
        del ys
    torch.cuda.empty_cache()
    return rep


def _accuracy(a, b):
    """benchmarks/utils.py:66-89."""
    a = np.asarray(a)
    b = np.asarray(b)
    if np.iscomplexobj(a) or np.iscomplexobj(b):
        a = np.stack([a.real, a.imag]).astype(np.float64)
        b = np.stack([b.real, b.imag]).astype(np.float64)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return {"max_abs_error": float(d.max()), "mean_abs_error": float(d.mean()),
            "correlation": float(np.corrcoef(a.ravel(), b.ravel())[0, 1])}


def published_rows(warmup=3, runs=10):
    """The six rows of the reference's benchmark tables, at its own shape and protocol: one clip of
    22 050 samples (benchmarks/utils.py:92-115), median of `runs` synchronised calls after `warmup`."""
    from oracle import audio_oracle as ao                     # checker only: accuracy beside the timing

    y_h = ao.bench_signal(22050, 22050, seed=42)
    y = torch.from_numpy(y_h).cuda()

    def lat(fn):
        for _ in range(warmup):
            fn()
            torch.cuda.synchronize()
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts)), float(np.mean(ts))

    rows = {"protocol": f"1 clip x 22050 samples @22.05 kHz, median (and mean) of {runs} synchronised calls after "
                        f"{warmup} warm-ups (benchmarks/utils.py:30-63); published = the reference's table on Apple "
                        "silicon (benchmarks/README.md:34-47,109-138), other hardware, not a baseline for `vs_baseline`"}
    S = ap.stft(y, n_fft=2048)
    med, mean = lat(lambda: ap.stft(y, n_fft=2048))
    rows["stft_n_fft2048"] = {"latency_ms": med, "mean_ms": mean, "published_ms": 0.29,
                              "accuracy": _accuracy(S.cpu().numpy(), ao.stft(y_h, n_fft=2048))}
    med, mean = lat(lambda: ap.istft(ap.stft(y, n_fft=2048), length=22050))
    yr = ap.istft(S, length=22050)
    rows["istft_round_trip_n_fft2048"] = {"latency_ms": med, "mean_ms": mean, "published_ms": 0.51,
                                          "accuracy": _accuracy(yr.cpu().numpy(), y_h)}
    M = ap.melspectrogram(y, sr=22050, n_fft=2048, n_mels=128)
    med, mean = lat(lambda: ap.melspectrogram(y, sr=22050, n_fft=2048, n_mels=128))
    rows["melspectrogram_n_mels128"] = {"latency_ms": med, "mean_ms": mean, "published_ms": 0.44,
                                        "accuracy": _accuracy(M.cpu().numpy(), ao.melspectrogram(y_h, sr=22050, n_fft=2048, n_mels=128))}
    C = ap.mfcc(y, sr=22050, n_mfcc=13)
    med, mean = lat(lambda: ap.mfcc(y, sr=22050, n_mfcc=13))
    rows["mfcc_n_mfcc13"] = {"latency_ms": med, "mean_ms": mean, "published_ms": 0.95,
                             "accuracy": _accuracy(C.cpu().numpy(), ao.mfcc(y_h, sr=22050, n_mfcc=13))}
    R = ap.resample_poly(y, 1, 2)
    med, mean = lat(lambda: ap.resample_poly(y, 1, 2))
    rows["resample_poly_integer_ratio"] = {"latency_ms": med, "mean_ms": mean, "published_ms": 0.45,
                                           "accuracy": _accuracy(R.cpu().numpy(), ao.resample_poly(y_h, 1, 2))}
    mag = ap.magnitude(S)
    med, mean = lat(lambda: ap.griffinlim(mag, n_iter=32, random_state=42, length=22050))
    rows["griffinlim_n_iter32"] = {"latency_ms": med, "mean_ms": mean, "published_ms": 15.2}
    return rows


def main():
    rep = run_configs(ramp_s=1.0)
    rep["published_rows"] = published_rows()
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
