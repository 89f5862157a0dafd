"""Steady-state timings of the SURVEY §8(f) rows on one MI355X (same protocol as tools/bench_configs.py:
rotating inputs resident in HBM, 1 s ramp-up, median of 5 back-to-back streams).  Prints a JSON report;
`frac_hbm` = algorithmic bytes / time / 8 TB/s.   usage: python tools/bench_features.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlx_audio_primitives_amd as ap  # noqa: E402
from bench_configs import N_ROT, rec, steady  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(7)
    B, L = 256, 220500
    ys = [torch.randn((B, L), device="cuda", generator=g) * 0.1 for _ in range(N_ROT)]
    T, F = 431, 1025
    rep = {"protocol": "3 rotating inputs, 1 s ramp-up, median of 5 back-to-back streams (HIP events)"}
    # rank 1: spectral statistics.  From audio: fused STFT kernel + one statistics pass over the complex spectrum
    ms = steady(lambda i: ap.spectral_features(ys[i % N_ROT], sr=22050))
    rep["spectral_features_from_audio_2048"] = rec(ms, B * T, (4 * 512 + 12) * B * T, unit="frames",
                                                   note="centroid + bandwidth + rolloff; algorithmic = samples in, 3 floats out")
    ms = steady(lambda i: ap.spectral_flatness(ys[i % N_ROT]))
    rep["spectral_flatness_from_audio_2048"] = rec(ms, B * T, (4 * 512 + 4) * B * T, unit="frames")
    Ss = [ap.magnitude(ap.stft(y)) for y in ys]
    ms = steady(lambda i: ap.spectral_features(S=Ss[i % N_ROT], sr=22050))
    rep["spectral_features_from_S_1025"] = rec(ms, B * T, (4 * F + 12) * B * T, unit="frames")
    ms = steady(lambda i: ap.spectral_rolloff(S=Ss[i % N_ROT], sr=22050))
    rep["spectral_rolloff_from_S_1025"] = rec(ms, B * T, (4 * F + 4) * B * T, unit="frames")
    del Ss
    ms = steady(lambda i: ap.zero_crossing_rate(ys[i % N_ROT]))
    rep["zero_crossing_rate_2048_512"] = rec(ms, B * T, (4 * 512 + 4) * B * T, unit="frames")
    ms = steady(lambda i: ap.rms(ys[i % N_ROT]))
    rep["rms_2048_512"] = rec(ms, B * T, (4 * 512 + 4) * B * T, unit="frames")
    # rank 2: emphasis filters, delta
    ms = steady(lambda i: ap.preemphasis(ys[i % N_ROT]))
    rep["preemphasis"] = rec(ms, B * L, 8 * B * L, unit="samples")
    ms = steady(lambda i: ap.deemphasis(ys[i % N_ROT]))
    rep["deemphasis"] = rec(ms, B * L, 8 * B * L, unit="samples")
    feats = [torch.randn((1024, 13, 313), device="cuda", generator=g) for _ in range(N_ROT)]
    ms = steady(lambda i: ap.delta(feats[i % N_ROT]))
    rep["delta_width9_mfcc13"] = rec(ms, 1024 * 13 * 313, 8 * 1024 * 13 * 313, unit="values")
    mels = [torch.rand((B, 128, T), device="cuda", generator=g) for _ in range(N_ROT)]
    ms = steady(lambda i: ap.delta(mels[i % N_ROT], order=2))
    rep["delta2_width9_mel128"] = rec(ms, B * 128 * T, 8 * B * 128 * T, unit="values")
    del feats, mels
    # rank 3: 16-bit PCM ingest fused into the mel kernel
    y16 = [(y * 32767 * 5).clamp(-32768, 32767).to(torch.int16) for y in ys]
    ms = steady(lambda i: ap.melspectrogram(y16[i % N_ROT], sr=22050, n_fft=2048, hop_length=512, n_mels=128))
    rep["mel2048_from_int16"] = rec(ms, B * T, (2 * 512 + 4 * 128) * B * T, unit="frames")
    ms = steady(lambda i: ap.pcm16_to_float(y16[i % N_ROT]))
    rep["pcm16_to_float"] = rec(ms, B * L, 6 * B * L, unit="samples")
    del y16
    # rank 4: other banks through the fused contraction, autocorrelation, streaming
    bark = ap.bark_filterbank(sr=22050, n_fft=2048, n_bands=24)
    ms = steady(lambda i: ap.filterbank_spectrogram(ys[i % N_ROT], bark, n_fft=2048, hop_length=512))
    rep["bark24_spectrogram_2048"] = rec(ms, B * T, (4 * 512 + 4 * 24) * B * T, unit="frames")
    ya = [y[:, :22050].contiguous() for y in ys]
    ms = steady(lambda i: ap.autocorrelation(ya[i % N_ROT], max_lag=2048))
    rep["autocorrelation_22050_lag2048"] = rec(ms, B * 22050, (4 * 22050 + 4 * 2048) * B, unit="samples")
    st = ap.StreamingSTFT(n_fft=2048, hop_length=512)
    chunk = 16384

    def stream_pass(i):
        y = ys[i % N_ROT]
        st.reset()
        for s in range(0, 8 * chunk, chunk):
            st.process(y[:, s:s + chunk])

    ms = steady(stream_pass, n_launch=5)
    rep["streaming_stft_8_chunks_of_16384"] = rec(ms, B * (8 * chunk // 512), (4 * 512 + 8 * F) * B * (8 * chunk // 512), unit="frames")
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
