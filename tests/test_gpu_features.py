"""GPU parity for the callers that sit directly on the STFT / on the frames (SURVEY.md §8f ranks
1-2): spectral centroid / bandwidth / rolloff / flatness, zero-crossing rate, frame, rms,
pre- / de-emphasis and delta — HIP kernels through the C ABI against the CPU oracle.  Mirrors the
reference's tests/test_features.py, tests/test_framing.py and tests/test_mfcc.py:127-164 (whose
librosa comparisons are restated in the oracle; librosa itself is absent: pinned by proxy through
the reference's formulas and SciPy — scipy.signal.savgol_filter / lfilter ARE the reference's
implementation of delta / deemphasis)."""

import numpy as np
import pytest

from oracle import audio_oracle as ao

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import mlx_audio_primitives_amd as ap  # noqa: E402


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ spectral features
@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (1024, 256), (512, 128), (400, 160)])
def test_spectral_centroid_bandwidth_flatness(random_signal, n_fft, hop):
    y = dev(random_signal)
    kw = dict(n_fft=n_fft, hop_length=hop)
    T = 1 + len(random_signal) // hop
    c = ap.spectral_centroid(y, sr=22050, **kw)
    assert c.shape == (1, T) and c.dtype == torch.float32
    np.testing.assert_allclose(host(c), ao.spectral_centroid(random_signal, sr=22050, **kw), rtol=1e-4, atol=1e-2)
    for p in (2.0, 1.0, 3.0):
        np.testing.assert_allclose(host(ap.spectral_bandwidth(y, sr=22050, p=p, **kw)),
                                   ao.spectral_bandwidth(random_signal, sr=22050, p=p, **kw), rtol=2e-4, atol=1e-2)
    np.testing.assert_allclose(host(ap.spectral_bandwidth(y, sr=22050, norm=False, **kw)),
                               ao.spectral_bandwidth(random_signal, sr=22050, norm=False, **kw), rtol=2e-4)
    for power in (2.0, 1.0):
        f = host(ap.spectral_flatness(y, power=power, **kw))
        assert f.shape == (1, T) and (f >= 0).all() and (f <= 1.0 + 1e-6).all()
        np.testing.assert_allclose(f, ao.spectral_flatness(random_signal, power=power, **kw), rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("n_bands,quantile,linear", [(6, 0.02, False), (4, 0.02, True), (8, 0.1, False),
                                                      (6, 0.5, True), (6, 1.0, False), (6, 0.0, True)])
def test_spectral_contrast(random_signal, chirp_signal, n_bands, quantile, linear):
    """reference features.py:445-595 / tests/test_features.py:259-291 (its librosa comparison is absent here:
    the oracle restates the band rule and is checked against a per-frame walk in test_oracle_golden)."""
    kw = dict(sr=22050, n_bands=n_bands, quantile=quantile, linear=linear)
    for sig in (random_signal, chirp_signal):
        got = ap.spectral_contrast(dev(sig), **kw)
        assert got.shape == (n_bands + 1, 1 + len(sig) // 512) and got.dtype == torch.float32
        S = host(ap.magnitude(ap.stft(dev(sig))))
        want = ao.spectral_contrast(S=S, **kw)
        if linear:
            np.testing.assert_allclose(host(got), want, rtol=2e-5, atol=1e-5 * float(np.abs(S).max()))
        else:
            np.testing.assert_allclose(host(got), want, rtol=1e-5, atol=2e-4)
        np.testing.assert_array_equal(host(ap.spectral_contrast(S=dev(S), **kw)), host(got))
    # the whole path from audio stays within the STFT's own tolerance of the oracle's
    np.testing.assert_allclose(host(ap.spectral_contrast(dev(random_signal), **kw)),
                               ao.spectral_contrast(random_signal, **kw), rtol=1e-3, atol=2e-2 if not linear else 2e-3)


def test_spectral_contrast_batched_other_grids_and_edges(batch_signals):
    y = dev(batch_signals)
    for kw in (dict(sr=16000, n_fft=512, hop_length=128, n_bands=5, fmin=100.0),
               dict(sr=16000, n_fft=400, hop_length=160, n_bands=6, fmin=200.0),       # top edge past Nyquist
               dict(sr=8000, n_fft=256, hop_length=64, n_bands=7, fmin=200.0),         # empty top bands
               dict(sr=22050, n_fft=1024, hop_length=256, n_bands=3, fmin=20.0, quantile=0.3)):
        S = host(ap.magnitude(ap.stft(y, n_fft=kw["n_fft"], hop_length=kw["hop_length"])))
        got = host(ap.spectral_contrast(y, **kw))
        assert got.shape == (batch_signals.shape[0], kw["n_bands"] + 1, S.shape[-1])
        np.testing.assert_allclose(got, ao.spectral_contrast(S=S, **kw), rtol=1e-5, atol=2e-4)
    # ties and constant columns: every selection order gives the same sums
    S = np.ones((2, 1025, 7), np.float32)
    S[:, ::3] = 0.25
    for linear in (True, False):
        np.testing.assert_allclose(host(ap.spectral_contrast(S=dev(S), linear=linear, quantile=0.2)),
                                   ao.spectral_contrast(S=S, linear=linear, quantile=0.2), rtol=1e-6, atol=1e-5)
    z = host(ap.spectral_contrast(S=torch.zeros(1025, 3).cuda()))
    assert z.shape == (7, 3) and (z == 0).all()                                        # both clamp to amin
    # custom bin centres, and the validation messages of the reference
    f = np.linspace(0, 4000.0, 1025).astype(np.float32)
    Sr = np.abs(np.random.default_rng(5).standard_normal((1025, 9))).astype(np.float32)
    np.testing.assert_allclose(host(ap.spectral_contrast(S=dev(Sr), freq=dev(f))), ao.spectral_contrast(S=Sr, freq=f),
                               rtol=1e-5, atol=2e-4)
    with pytest.raises(ValueError, match="n_bands must be positive"):
        ap.spectral_contrast(S=dev(Sr), n_bands=0)
    with pytest.raises(ValueError, match="quantile must be"):
        ap.spectral_contrast(S=dev(Sr), quantile=1.5)
    with pytest.raises(ValueError, match="Either y"):
        ap.spectral_contrast()
    assert ap.spectral_contrast(S=torch.zeros(2, 1025, 0).cuda()).shape == (2, 7, 0)


def _rolloff_agrees(got, want, freq, frac=0.02):
    """The rolloff is a bin frequency: where the float32 running sum meets the threshold within
    rounding, another summation order may pick the neighbouring bin.  Allow that for < 2 % of frames."""
    assert got.shape == want.shape
    step = freq[1] - freq[0]
    diff = np.abs(got - want)
    assert (diff <= step * 1.001).all(), diff.max()
    assert (diff > 0).mean() <= frac, (diff > 0).mean()


@pytest.mark.parametrize("roll_percent", [0.85, 0.5, 0.95, 0.1, 1.0, 0.0])
def test_spectral_rolloff(random_signal, chirp_signal, roll_percent):
    freq = ao.fft_frequencies(22050, 2048)
    for sig in (random_signal, chirp_signal):
        got = host(ap.spectral_rolloff(dev(sig), sr=22050, roll_percent=roll_percent))
        if roll_percent == 1.0:
            # "running sum >= total" is decided by the last bits of a float32 sum of 1025 terms: any
            # summation order (the reference's is MLX's parallel scan) may stop a few bins early.
            # Well defined: not below the 99.99 % point, not above Nyquist.
            lo = ao.spectral_rolloff(sig, sr=22050, roll_percent=0.9999)
            assert (got >= lo - 11.0).all() and (got <= freq[-1]).all()
            continue
        _rolloff_agrees(got, ao.spectral_rolloff(sig, sr=22050, roll_percent=roll_percent), freq)
        assert (got >= 0).all() and (got <= 22050 / 2).all()                  # tests/test_features.py:186-193
    with pytest.raises(ValueError, match="roll_percent must be"):
        ap.spectral_rolloff(dev(random_signal), roll_percent=1.5)


def test_spectral_features_from_spectrogram_batch_and_given_centroid(batch_signals):
    y = batch_signals[:, :12000]
    S = ao.magnitude(ao.stft(y))                                            # (4, 1025, T)
    Sd = dev(S)
    c = ap.spectral_centroid(S=Sd, sr=22050)
    assert c.shape == (4, 1, S.shape[-1])
    np.testing.assert_allclose(host(c), ao.spectral_centroid(S=S, sr=22050), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(host(ap.spectral_centroid(dev(y), sr=22050)), host(c), rtol=1e-4, atol=1e-2)
    # unbatched S, custom bin centres, bandwidth around a supplied centroid
    fr = np.linspace(0, 1.0, 1025).astype(np.float32)
    np.testing.assert_allclose(host(ap.spectral_centroid(S=Sd[1], freq=dev(fr))), ao.spectral_centroid(S=S[1], freq=fr),
                               rtol=1e-4, atol=1e-6)
    cen = np.full((4, 1, S.shape[-1]), 3000.0, np.float32)
    np.testing.assert_allclose(host(ap.spectral_bandwidth(S=Sd, sr=22050, centroid=dev(cen))),
                               ao.spectral_bandwidth(S=S, sr=22050, centroid=cen), rtol=2e-4)
    np.testing.assert_allclose(host(ap.spectral_flatness(S=dev(S ** 2))), ao.spectral_flatness(S=S ** 2), rtol=2e-4, atol=1e-6)
    # all three from one pass == the three separate calls
    one = ap.spectral_features(dev(y), sr=22050)
    assert torch.equal(one["centroid"], ap.spectral_centroid(dev(y), sr=22050))
    assert torch.equal(one["bandwidth"], ap.spectral_bandwidth(dev(y), sr=22050))
    assert torch.equal(one["rolloff"], ap.spectral_rolloff(dev(y), sr=22050))
    with pytest.raises(ValueError, match="Either y"):
        ap.spectral_centroid()


def test_spectral_feature_properties():
    """tests/test_features.py:89-101,236-259: a chirp's centroid stays inside the band; white noise is
    flat, a sine is not; a pure tone's centroid sits at the tone."""
    sr = 22050
    t = np.arange(sr, dtype=np.float32) / sr
    tone = np.sin(2 * np.pi * 2000.0 * t).astype(np.float32)
    c = host(ap.spectral_centroid(dev(tone), sr=sr))[0, 2:-2]
    assert np.abs(c - 2000.0).max() < 25.0
    noise = np.random.default_rng(1).standard_normal(sr).astype(np.float32)
    assert host(ap.spectral_flatness(dev(noise))).mean() > 0.3
    assert host(ap.spectral_flatness(dev(tone)))[0, 2:-2].mean() < 0.01
    assert host(ap.spectral_bandwidth(dev(tone), sr=sr))[0, 2:-2].max() < 400.0


def test_spectral_stats_large_batch_matches_oracle_subsample():
    """Headline-shaped batch (64 x 5 s): every tile shape of the statistics kernel incl. the ragged
    last tile (T = 216 = 6 x 32 + 24)."""
    g = torch.Generator(device="cuda").manual_seed(6)
    y = torch.randn((64, 110250), device="cuda", generator=g)
    f = ap.spectral_features(y, sr=22050)
    assert f["centroid"].shape == (64, 1, 216)
    idx = [0, 31, 63]
    yh = host(y[idx])
    np.testing.assert_allclose(host(f["centroid"][idx]), ao.spectral_centroid(yh, sr=22050), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(host(f["bandwidth"][idx]), ao.spectral_bandwidth(yh, sr=22050), rtol=2e-4, atol=1e-2)
    _rolloff_agrees(host(f["rolloff"][idx]), ao.spectral_rolloff(yh, sr=22050), ao.fft_frequencies(22050, 2048))


# ------------------------------------------------------------------ frame / rms / zcr
@pytest.mark.parametrize("frame_length,hop", [(2048, 512), (1024, 256), (512, 128), (400, 160), (100, 33), (3, 1)])
def test_rms_and_zcr(random_signal, frame_length, hop):
    y = dev(random_signal)
    for center in (True, False):
        for pad_mode in ("constant", "edge"):
            kw = dict(frame_length=frame_length, hop_length=hop, center=center, pad_mode=pad_mode)
            r = ap.rms(y, **kw)
            want = ao.rms(random_signal, **kw)
            assert r.shape == want.shape
            np.testing.assert_allclose(host(r), want, rtol=1e-5, atol=1e-7)
            z = ap.zero_crossing_rate(y, **kw)
            np.testing.assert_array_equal(host(z), ao.zero_crossing_rate(random_signal, **kw))   # counts: exact


def test_rms_zcr_batch_edges_and_errors(batch_signals):
    y = dev(batch_signals)
    assert ap.rms(y).shape == (4, 1, 44) and ap.zero_crossing_rate(y).shape == (4, 1, 44)
    np.testing.assert_allclose(host(ap.rms(y)), ao.rms(batch_signals), rtol=1e-5)
    np.testing.assert_array_equal(host(ap.zero_crossing_rate(y)), ao.zero_crossing_rate(batch_signals))
    # a signal with exact zeros and negative zeros: (x >= 0) is the sign test (features.py:607-616)
    x = np.array([0.0, -0.0, 1.0, -1.0, 0.0, 0.0, -2.0, 3.0] * 40, np.float32)
    np.testing.assert_array_equal(host(ap.zero_crossing_rate(dev(x), frame_length=16, hop_length=8)),
                                  ao.zero_crossing_rate(x, frame_length=16, hop_length=8))
    # high-frequency content crosses often, a constant never (tests/test_features.py:347-360)
    alt = np.tile(np.array([1.0, -1.0], np.float32), 4000)
    assert host(ap.zero_crossing_rate(dev(alt), center=False)).min() > 0.99
    assert host(ap.zero_crossing_rate(dev(np.ones(8000, np.float32)))).max() == 0.0
    with pytest.raises(ValueError, match="must be positive"):
        ap.rms(y, frame_length=0)
    with pytest.raises(ValueError, match="Unknown pad_mode"):
        ap.rms(y, pad_mode="reflect")
    with pytest.raises(ValueError, match="must be >= frame_length"):
        ap.rms(dev(np.zeros(10, np.float32)), frame_length=64, center=False)


def test_frame_api(random_signal, batch_signals):
    f = ap.frame(dev(random_signal), 2048, 512)
    np.testing.assert_array_equal(host(f), ao.frame(random_signal, 2048, 512))
    assert ap.frame(dev(batch_signals), 1024, 256).shape == (4, 1 + (22050 - 1024) // 256, 1024)
    with pytest.raises(ValueError, match="axis must be -1"):
        ap.frame(dev(random_signal), 2048, 512, axis=0)
    with pytest.raises(ValueError, match="must be positive"):
        ap.frame(dev(random_signal), 0, 512)


# ------------------------------------------------------------------ pre- / de-emphasis
@pytest.mark.parametrize("coef", [0.97, 0.5, 0.0, 1.0])
def test_preemphasis_deemphasis(random_signal, batch_signals, coef):
    # 22 050 / 5 / 4 097 samples: the one-sample kernel; 4 096 / 22 048 (x 3 clips): the 16-byte kernel
    for sig in (random_signal, batch_signals, random_signal[:5], random_signal[:4097], random_signal[:4096],
                np.ascontiguousarray(batch_signals[:3, :22048])):
        p, zf = ap.preemphasis(dev(sig), coef=coef, return_zf=True)
        pw, zfw = ao.preemphasis(sig, coef=coef, return_zf=True)
        np.testing.assert_allclose(host(p), pw, rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(host(zf), zfw)
        d, dzf = ap.deemphasis(p, coef=coef, return_zf=True)
        dw, dzfw = ao.deemphasis(pw, coef=coef, return_zf=True)
        np.testing.assert_allclose(host(d), dw, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(host(dzf), dzfw, rtol=1e-4, atol=2e-5)
        if coef < 1.0:
            np.testing.assert_allclose(host(d), sig, rtol=1e-4, atol=5e-5)     # round trip, tests/test_framing.py:198-209


def test_emphasis_with_given_state_and_errors(batch_signals):
    y = batch_signals[:, :3000]
    for zi in (0.25, np.array([0.1, -0.2, 0.3, 0.0], np.float32)):
        np.testing.assert_allclose(host(ap.preemphasis(dev(y), zi=zi)), ao.preemphasis(y, zi=zi), rtol=1e-6, atol=1e-6)
        d, zf = ap.deemphasis(dev(y), zi=zi, return_zf=True)
        dw, zfw = ao.deemphasis(y, zi=zi, return_zf=True)
        np.testing.assert_allclose(host(d), dw, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(host(zf), zfw, rtol=1e-4, atol=2e-5)
    with pytest.raises(ValueError, match=r"coef must be in \[0, 1\]"):
        ap.preemphasis(dev(y), coef=1.5)
    with pytest.raises(ValueError, match=r"coef must be in \[0, 1\]"):
        ap.deemphasis(dev(y), coef=-0.1)


@pytest.mark.parametrize("coef", [0.97, 0.999, 1.0])
@pytest.mark.parametrize("zi", [None, 0.3])
def test_deemphasis_chunked_long_clips(coef, zi):
    """Clips longer than 16 384 samples run as chunks on their own workgroups (end states, then the
    chunks from their composed entering states): 7 chunks, the last one ragged; a slowly decaying and a
    non-decaying (coef = 1) recursion carry state across every chunk border."""
    rng = np.random.default_rng(17)
    y = (rng.standard_normal((3, 100001)) * 0.1).astype(np.float32)
    d, zf = ap.deemphasis(dev(y), coef=coef, zi=zi, return_zf=True)
    dw, zfw = ao.deemphasis(y, coef=coef, zi=zi, return_zf=True)
    scale = float(np.abs(dw).max())
    np.testing.assert_allclose(host(d), dw, rtol=1e-4, atol=2e-5 * max(1.0, scale))
    np.testing.assert_allclose(host(zf), zfw, rtol=1e-4, atol=2e-5 * max(1.0, scale))


# ------------------------------------------------------------------ delta
@pytest.mark.parametrize("width,order", [(9, 1), (9, 2), (5, 1), (3, 1), (7, 2)])
def test_delta_matches_savgol(random_signal, width, order):
    M = ao.mfcc(random_signal, n_mfcc=13)                                    # (13, 44)
    d = ap.delta(dev(M), width=width, order=order)
    assert d.shape == M.shape
    np.testing.assert_allclose(host(d), ao.delta(M, width=width, order=order), rtol=1e-4, atol=1e-4)
    for mode in ("nearest", "mirror", "constant", "wrap"):
        np.testing.assert_allclose(host(ap.delta(dev(M), width=width, order=order, mode=mode)),
                                   ao.delta(M, width=width, order=order, mode=mode), rtol=1e-4, atol=1e-4, err_msg=mode)
    # along another axis, batched input
    Mb = np.stack([M, 2 * M, -M])
    np.testing.assert_allclose(host(ap.delta(dev(Mb), width=width, order=order, axis=1)),
                               ao.delta(Mb, width=width, order=order, axis=1), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(ap.delta(dev(Mb), width=width, order=order)),
                               ao.delta(Mb, width=width, order=order), rtol=1e-4, atol=1e-4)


def test_delta_errors_and_full_frontend(random_signal):
    M = dev(ao.mfcc(random_signal, n_mfcc=13))
    with pytest.raises(ValueError, match="width must be >= 3"):
        ap.delta(M, width=1)
    with pytest.raises(ValueError, match="width must be odd"):
        ap.delta(M, width=4)
    with pytest.raises(ValueError, match="cannot exceed"):
        ap.delta(M[:, :5], width=9)
    with pytest.raises(ValueError, match="must be positive"):
        ap.delta(M, order=0)
    # the ASR front end of BASELINE config 4 completed on the device: preemphasis -> mfcc -> delta, delta-delta
    y = random_signal
    feats = torch.cat([ap.mfcc(ap.preemphasis(dev(y)), n_mfcc=13), ap.delta(ap.mfcc(ap.preemphasis(dev(y)), n_mfcc=13)),
                       ap.delta(ap.mfcc(ap.preemphasis(dev(y)), n_mfcc=13), order=2)])
    m = ao.mfcc(ao.preemphasis(y), n_mfcc=13)
    want = np.concatenate([m, ao.delta(m), ao.delta(m, order=2)])
    assert feats.shape == (39, 44)
    np.testing.assert_allclose(host(feats), want, rtol=1e-3, atol=5e-3)


# ------------------------------------------------------------------ fused statistics from audio (n_fft = 2048)
@pytest.mark.parametrize("B,L,hop,center", [(300, 9000, 512, True), (5, 40001, 256, True), (3, 30000, 512, False),
                                            (2, 2048, 2048, False)])
def test_spectral_statistics_fused_from_audio_vs_two_kernel_route(B, L, hop, center):
    """ap_spec2048_run_kernel (samples -> statistics, the spectrum never written) against the STFT kernel
    + statistics kernel route (AP_SPEC_TWO_KERNELS) and the oracle; more frames than workgroups, a hop the
    register rotation does not serve, no centring, a single frame."""
    import os
    rng = np.random.default_rng(B + L)
    t = np.arange(L) / 22050.0
    y = (0.3 * np.sin(2 * np.pi * (300.0 + 40.0 * np.arange(B))[:, None] * t[None]) +
         0.05 * rng.standard_normal((B, L))).astype(np.float32)
    y[0, : L // 2] = 0.0
    kw = dict(sr=22050, n_fft=2048, hop_length=hop, center=center)
    yd = dev(y)
    fused = ap.spectral_features(yd, **kw)
    flat = ap.spectral_flatness(yd, n_fft=2048, hop_length=hop, center=center)
    bw3 = ap.spectral_bandwidth(yd, p=3.0, norm=False, **kw)
    flat15 = ap.spectral_flatness(yd, n_fft=2048, hop_length=hop, center=center, power=1.5, amin=1e-6)
    os.environ["AP_SPEC_TWO_KERNELS"] = "1"
    try:
        two = ap.spectral_features(yd, **kw)
        flat2 = ap.spectral_flatness(yd, n_fft=2048, hop_length=hop, center=center)
        bw32 = ap.spectral_bandwidth(yd, p=3.0, norm=False, **kw)
        flat152 = ap.spectral_flatness(yd, n_fft=2048, hop_length=hop, center=center, power=1.5, amin=1e-6)
    finally:
        del os.environ["AP_SPEC_TWO_KERNELS"]
    bin_hz = 22050 / 2048
    for k in ("centroid", "bandwidth"):
        assert fused[k].shape == two[k].shape
        np.testing.assert_allclose(host(fused[k]), host(two[k]), rtol=2e-4, atol=2e-2)
    off = np.abs(host(fused["rolloff"]) - host(two["rolloff"])) / bin_hz
    assert off.max() <= 1.001 and (off > 0.5).mean() < 0.02
    np.testing.assert_allclose(host(flat), host(flat2), rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(host(bw3), host(bw32), rtol=5e-4, atol=1e-1)
    np.testing.assert_allclose(host(flat15), host(flat152), rtol=2e-3, atol=1e-7)
    for b in sorted(set([0, B // 2, B - 1])):
        np.testing.assert_allclose(host(fused["centroid"])[b], ao.spectral_centroid(y[b], **kw), rtol=2e-4, atol=2e-2)
        np.testing.assert_allclose(host(fused["bandwidth"])[b], ao.spectral_bandwidth(y[b], **kw), rtol=2e-4, atol=2e-2)
        np.testing.assert_allclose(host(flat)[b], ao.spectral_flatness(y[b], n_fft=2048, hop_length=hop, center=center),
                                   rtol=2e-3, atol=1e-7)
        offo = np.abs(host(fused["rolloff"])[b] - ao.spectral_rolloff(y[b], **kw)) / bin_hz
        assert offo.max() <= 1.001 and (offo > 0.5).mean() < 0.02


def test_spectral_statistics_fused_custom_freq_and_1d():
    rng = np.random.default_rng(3)
    y = rng.standard_normal(20000).astype(np.float32)
    freq = (np.linspace(0, 1, 1025) ** 2 * 8000).astype(np.float32)
    got = ap.spectral_centroid(dev(y), freq=freq)
    assert got.shape == (1, 40)
    np.testing.assert_allclose(host(got), ao.spectral_centroid(y, freq=freq), rtol=2e-4, atol=2e-2)
    with pytest.raises(ValueError, match="freq must be 1D"):
        ap.spectral_centroid(dev(y), freq=freq[:100])


@pytest.mark.parametrize("frame_length,hop,L,B,center,pad_mode", [
    (2048, 512, 50001, 3, True, "edge"), (2048, 512, 50001, 3, True, "constant"), (1024, 256, 9000, 70, True, "edge"),
    (400, 160, 16001, 2, True, "edge"), (400, 100, 3000, 2, False, "constant"), (2048, 2048, 30000, 2, True, "edge"),
    (4096, 1024, 20000, 1, True, "constant"), (512, 128, 700, 5, True, "edge"), (64, 4, 500, 2, True, "edge"),
])
def test_rms_zcr_block_kernel_vs_span_kernel_and_oracle(frame_length, hop, L, B, center, pad_mode):
    """frame_length = m hop: the block-partial kernel (every sample read once) against the LDS-span kernel
    (AP_FRAME_STATS_SPAN) and the oracle; zero-crossing counts are exact."""
    import os
    rng = np.random.default_rng(frame_length + hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    y[0, 100:400] = 0.0                                   # exact zeros: (x >= 0) sign tests
    kw = dict(frame_length=frame_length, hop_length=hop, center=center, pad_mode=pad_mode)
    r, z = host(ap.rms(dev(y), **kw)), host(ap.zero_crossing_rate(dev(y), **kw))
    os.environ["AP_FRAME_STATS_SPAN"] = "1"
    try:
        r2, z2 = host(ap.rms(dev(y), **kw)), host(ap.zero_crossing_rate(dev(y), **kw))
    finally:
        del os.environ["AP_FRAME_STATS_SPAN"]
    np.testing.assert_allclose(r, r2, rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(z, z2)
    np.testing.assert_allclose(r, ao.rms(y, **kw), rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(z, ao.zero_crossing_rate(y, **kw))


# ------------------------------------------------------------------ pitch_detect_acf / periodicity
def _tones(B, L, sr, freqs, noise=0.02, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(L) / sr
    y = np.stack([np.sin(2 * np.pi * f * t) + 0.4 * np.sin(2 * np.pi * 2 * f * t) for f in freqs[:B]])
    return (y + noise * rng.standard_normal((B, L))).astype(np.float32)


@pytest.mark.parametrize("kw", [dict(), dict(fmin=80.0, fmax=500.0, frame_length=1024, hop_length=256),
                                dict(center=False, threshold=0.3), dict(sr=16000, fmin=60.0, fmax=400.0)])
def test_pitch_detect_acf_and_periodicity(kw):
    """reference pitch.py:118-369 (tests/test_pitch.py: a 220 Hz tone is found within a few Hz; silence is
    unvoiced): frames + device autocorrelation + peak pass against the oracle's restatement."""
    sr = kw.get("sr", 22050)
    y = _tones(4, 20000, sr, [110.0, 220.0, 330.0, 147.0])
    y[3, 5000:12000] = 0.0                                    # a silent stretch: unvoiced frames, periodicity 0
    f0, voiced = ap.pitch_detect_acf(dev(y), **kw)
    wf0, wv = ao.pitch_detect_acf(y, **kw)
    assert f0.shape == wf0.shape and voiced.dtype == torch.bool
    agree = host(voiced) == wv
    assert agree.mean() > 0.995                               # float32 vs float64 r at the threshold
    both = agree & wv
    close = np.isclose(host(f0)[both], wf0[both], rtol=1e-6)
    assert close.mean() > 0.995                               # a tie between neighbouring lags may flip
    pk = {k: v for k, v in kw.items() if k != "threshold"}
    p = host(ap.periodicity(dev(y), **pk))
    np.testing.assert_allclose(p, ao.periodicity(y, **pk), rtol=1e-4, atol=2e-5)
    assert p.shape == (4, 1, f0.shape[1])
    # the tone clips are found
    med = np.median(host(f0)[1][host(voiced)[1]])
    assert abs(med - 220.0) < 4.0
    f1, v1 = ap.pitch_detect_acf(dev(y[0]), **kw)
    assert f1.shape == (f0.shape[1],)
    np.testing.assert_array_equal(host(f1), host(f0)[0])
    with pytest.raises(ValueError, match="must be less than fmax"):
        ap.pitch_detect_acf(dev(y), fmin=500.0, fmax=100.0)
