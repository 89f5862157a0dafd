"""GPU parity for SURVEY.md §8f ranks 3-4: 16-bit PCM ingest fused ahead of the mel kernel,
streaming (chunked) STFT / mel, linear / Bark filterbanks through the fused contraction kernels,
autocorrelation."""

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


# ------------------------------------------------------------------ int16 ingest
def _pcm(shape, seed):
    rng = np.random.default_rng(seed)
    return np.clip(rng.standard_normal(shape) * 6000.0, -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("kw", [
    dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128),                    # fused into the run kernel
    dict(sr=16000, n_fft=2048, hop_length=512, n_mels=80, center=False),       # fused, no centring
    dict(sr=22050, n_fft=2048, hop_length=300, n_mels=64),                     # fused, full reload per frame
    dict(sr=22050, n_fft=2048, hop_length=300, n_mels=128),                    # fused; a stretch starts on a frame that straddles sample 0
    dict(sr=16000, n_fft=400, hop_length=160, n_mels=80),                      # conversion pass + ct engine
    dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128, pad_mode="reflect"),  # conversion pass + tile kernel
    dict(sr=22050, n_fft=1024, hop_length=256, n_mels=64, power=1.0),
])
def test_melspectrogram_from_int16_pcm(kw):
    x = _pcm((5, 30000), 16)
    want = ao.melspectrogram(x.astype(np.float32) / 32768.0, **kw)
    got = ap.melspectrogram(dev(x), **kw)
    assert got.shape == want.shape and got.dtype == torch.float32
    np.testing.assert_allclose(host(got), want, rtol=1e-4, atol=1e-6)
    # identical to converting first (same kernels or the fused conversion: within float32 rounding)
    via_float = ap.melspectrogram(ap.pcm16_to_float(dev(x)), **kw)
    np.testing.assert_allclose(host(got), host(via_float), rtol=2e-5, atol=1e-7)
    # numpy int16 input and a 1-D clip
    np.testing.assert_allclose(host(ap.melspectrogram(x[0], **kw)), want[0], rtol=1e-4, atol=1e-6)


def test_pcm16_edges_and_mfcc():
    x = _pcm((3, 20001), 7)                                   # odd length: not fusable, conversion pass
    kw = dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128)
    np.testing.assert_allclose(host(ap.melspectrogram(dev(x), **kw)),
                               ao.melspectrogram(x.astype(np.float32) / 32768.0, **kw), rtol=1e-4, atol=1e-6)
    full = np.array([[-32768, 32767, 0, 1, -1] * 2000], np.int16)
    np.testing.assert_array_equal(host(ap.pcm16_to_float(dev(full))), full.astype(np.float32) / 32768.0)
    np.testing.assert_array_equal(host(ap.pcm16_to_float(dev(full[0, :7]))), full[0, :7].astype(np.float32) / 32768.0)
    x2 = _pcm((4, 32000), 8)
    got = ap.mfcc(dev(x2), sr=16000, n_mfcc=13, n_fft=2048, hop_length=512, n_mels=128)
    want = ao.mfcc(x2.astype(np.float32) / 32768.0, sr=16000, n_mfcc=13, n_fft=2048, hop_length=512, n_mels=128)
    np.testing.assert_allclose(host(got), want, rtol=1e-4, atol=2e-3)


def test_pcm16_headline_shape_full_scale_property():
    """Headline-shaped batch: scaling the PCM by 2 (exact in int16 here) scales the power mel by 4."""
    g = torch.Generator(device="cuda").manual_seed(9)
    x = (torch.randn((64, 220500), device="cuda", generator=g) * 3000).clamp(-16000, 16000).to(torch.int16)
    kw = dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128)
    M = ap.melspectrogram(x, **kw)
    assert M.shape == (64, 128, 431)
    assert torch.equal(ap.melspectrogram(x * 2, **kw), M * 4.0)
    np.testing.assert_allclose(host(M[[0, 63]]), ao.melspectrogram(host(x[[0, 63]]).astype(np.float32) / 32768.0, **kw),
                               rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------ streaming
@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (1024, 256), (400, 160), (512, 500)])
def test_streaming_stft_equals_offline(n_fft, hop):
    rng = np.random.default_rng(n_fft)
    y = rng.standard_normal((3, 40000)).astype(np.float32)
    whole = ap.stft(dev(y), n_fft=n_fft, hop_length=hop, center=False)
    st = ap.StreamingSTFT(n_fft=n_fft, hop_length=hop)
    cuts = [0, 100, 2500, 2501, 9000, 9000 + n_fft, 25000, 40000]           # ragged chunks, one shorter than a hop
    parts = [st.process(dev(y[:, a:b])) for a, b in zip(cuts[:-1], cuts[1:])]
    got = torch.cat(parts, dim=-1)
    assert got.shape == whole.shape and st.frames_emitted == whole.shape[-1]
    assert torch.equal(torch.view_as_real(got), torch.view_as_real(whole))   # same kernels, same bits
    assert st.flush().shape[-1] == 0
    np.testing.assert_allclose(host(got), ao.stft(y, n_fft=n_fft, hop_length=hop, center=False), rtol=1e-4, atol=1e-4)


def test_streaming_mel_centered_equals_offline():
    rng = np.random.default_rng(3)
    y = rng.standard_normal(50000).astype(np.float32)
    kw = dict(sr=22050, n_mels=80)
    whole = ap.melspectrogram(dev(y), n_fft=2048, hop_length=512, center=True, **kw)
    st = ap.StreamingSTFT(n_fft=2048, hop_length=512, center=True, **kw)
    parts = [st.process(dev(y[a:a + 7000])) for a in range(0, 50000, 7000)] + [st.flush()]
    got = torch.cat(parts, dim=-1)
    assert got.shape == whole.shape
    assert torch.equal(got, whole)
    st.reset()
    assert st.process(dev(y[:100])).shape == (80, 0)


# ------------------------------------------------------------------ linear / Bark banks
@pytest.mark.parametrize("which", ["bark", "bark-t", "linear"])
def test_bark_and_linear_banks_through_the_fused_kernels(random_signal, which):
    if which == "linear":
        fb = ap.linear_filterbank(22050, 2048, 64)
        want_fb = ao.linear_filterbank(22050, 2048, 64)
    else:
        formula = "traunmuller" if which == "bark-t" else "zwicker"
        fb = ap.bark_filterbank(22050, 2048, 24, formula=formula)
        want_fb = ao.bark_filterbank(22050, 2048, 24, formula=formula)
    assert fb.is_cuda
    np.testing.assert_array_equal(host(fb), want_fb)
    got = ap.filterbank_spectrogram(dev(random_signal), fb, n_fft=2048, hop_length=512)
    S = ao.magnitude(ao.stft(random_signal)).astype(np.float64) ** 2
    want = (want_fb.astype(np.float64) @ S).astype(np.float32)
    np.testing.assert_allclose(host(got), want, rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError, match="cannot exceed Nyquist"):
        ap.bark_filterbank(22050, 2048, 24, fmax=20000.0)
    with pytest.raises(ValueError, match="Unknown formula"):
        ap.hz_to_bark(np.array([100.0]), formula="x")


# ------------------------------------------------------------------ autocorrelation
@pytest.mark.parametrize("n", [22050, 4096, 2205, 1, 5, 70001])
def test_autocorrelation(n):
    rng = np.random.default_rng(n)
    y = rng.standard_normal((3, n)).astype(np.float32) + 0.3
    for kw in (dict(), dict(normalize=False), dict(center=False), dict(max_lag=min(500, n))):
        got = ap.autocorrelation(dev(y), **kw)
        want = ao.autocorrelation(y, **kw)
        assert got.shape == want.shape
        scale = np.abs(want).max()
        np.testing.assert_allclose(host(got), want, rtol=1e-4, atol=1e-4 * max(scale, 1.0))
    r = host(ap.autocorrelation(dev(y[0])))
    assert r.shape == (n,) and (n == 1 or np.isclose(r[0], 1.0, rtol=1e-5))


def test_autocorrelation_sine_peak():
    """tests/test_pitch.py:56-76: a 440 Hz sine peaks at lag sr / 440."""
    sr = 22050
    t = np.linspace(0, 0.1, int(sr * 0.1), dtype=np.float32)
    r = host(ap.autocorrelation(dev(np.sin(2 * np.pi * 440 * t).astype(np.float32)), max_lag=1000))
    assert r.shape == (1000,)
    lo, hi = int(sr / 440 * 0.8), int(sr / 440 * 1.2)
    assert abs(lo + int(np.argmax(r[lo:hi])) - sr / 440) < 5


# ------------------------------------------------------------------ resample_poly padtypes
_EXT = ["line", "symmetric", "reflect", "edge", "wrap", "smooth", "antisymmetric", "antireflect"]


@pytest.mark.parametrize("mode", ["constant"] + _EXT)
def test_extend_matches_scipy_bit_for_bit(mode):
    """ap_extend_f32 against scipy.signal.upfirdn's own extension (its _pad_test hook) and the oracle."""
    import ctypes
    from scipy.signal._upfirdn_apply import _pad_test
    from mlx_audio_primitives_amd import _extension as ext
    from mlx_audio_primitives_amd.resample import _EXT_MODES
    rng = np.random.default_rng(11)
    for B, L, n_ext in ((3, 1000, 70), (2, 5, 19), (1, 2, 7), (4, 333, 333)):
        x = (rng.standard_normal((B, L)) + 0.3).astype(np.float32)
        xd = dev(x)
        out = torch.empty((B, L + 2 * n_ext), dtype=torch.float32, device="cuda")
        ext.check(ext.dlib(xd.device).ap_extend_f32(ext.ptr(xd), B, L, n_ext, _EXT_MODES[mode], ext.ptr(out),
                                                    ext.stream_ptr(xd.device)))
        got = host(out)
        for b in range(B):
            np.testing.assert_array_equal(got[b], _pad_test(x[b], npre=n_ext, npost=n_ext, mode=mode))
            np.testing.assert_array_equal(got[b], ao.upfirdn_extend(x[b], n_ext, mode))


@pytest.mark.parametrize("padtype", _EXT)
@pytest.mark.parametrize("up,down", [(1, 3), (2, 1), (3, 2), (160, 147), (1, 8)])
def test_resample_poly_extension_padtypes_bit_exact(padtype, up, down):
    """Reference resample.py:279-281 forwards padtype to scipy.signal.resample_poly: same float32 taps,
    same extension samples, same accumulation order -> identical bits."""
    import scipy.signal
    rng = np.random.default_rng(up * 100 + down)
    x = (rng.standard_normal((3, 4001)) + 0.5).astype(np.float32)
    want = scipy.signal.resample_poly(x, up, down, axis=-1, padtype=padtype).astype(np.float32)
    got = host(ap.resample_poly(dev(x), up, down, padtype=padtype))
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(ao.resample_poly(x, up, down, padtype=padtype), want)


@pytest.mark.parametrize("padtype", ["mean", "median", "minimum", "maximum"])
def test_resample_poly_background_padtypes(padtype):
    import scipy.signal
    rng = np.random.default_rng(5)
    for L in (4000, 4001):
        x = (rng.standard_normal((4, L)) + 3.0).astype(np.float32)
        want = scipy.signal.resample_poly(x, 1, 3, axis=-1, padtype=padtype).astype(np.float32)
        got = host(ap.resample_poly(dev(x), 1, 3, padtype=padtype))
        if padtype == "mean":            # the row mean is summed in another order than np.mean's pairwise sum
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=2e-6)
        else:
            np.testing.assert_array_equal(got, want)


def test_resample_poly_padtype_errors_and_short_signals():
    import scipy.signal
    x = np.arange(12, dtype=np.float32).reshape(2, 6)
    with pytest.raises(ValueError, match="padtype must be one of"):
        ap.resample_poly(dev(x), 1, 2, padtype="nearest")
    with pytest.raises(ValueError, match="at least two samples"):
        ap.resample_poly(dev(x[:, :1]), 1, 2, padtype="line")
    for padtype in _EXT:                  # the extension is many signal lengths deep
        want = scipy.signal.resample_poly(x, 3, 2, axis=-1, padtype=padtype).astype(np.float32)
        np.testing.assert_array_equal(host(ap.resample_poly(dev(x), 3, 2, padtype=padtype)), want)


@pytest.mark.parametrize("B", [5, 7, 16])
def test_int16_first_samples_of_a_clip_at_a_stretch_start(B):
    """Regression: 4-byte bounds-checked loads whose offset the compiler had split into a negative register
    part and an instruction immediate summing to 0 or 4 came back as 0, so the frame a wave STARTS its
    stretch with lost the first two samples of its clip (only when that frame straddles sample 0)."""
    x = np.zeros((B, 30000), np.int16)
    x[:, 0] = 10000                                # an impulse on the very first sample of every clip
    x[:, 2] = -7000
    kw = dict(sr=22050, n_fft=2048, hop_length=300, n_mels=128)
    got = host(ap.melspectrogram(dev(x), **kw))
    want = ao.melspectrogram(x.astype(np.float32) / 32768.0, **kw)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("n_fft,hop,n_mels", [(2048, 512, 128), (2048, 300, 128), (2048, 256, 80), (1024, 256, 80), (1024, 200, 64),
                                              (512, 128, 64), (512, 100, 40), (400, 160, 80), (400, 100, 80), (256, 64, 32)])
@pytest.mark.parametrize("B", [7, 64])
def test_first_and_last_samples_of_every_clip_reach_their_frames(n_fft, hop, n_mels, B):
    """Impulses on the first three and the last three samples of every clip: every frame that overlaps a clip end
    must see them (bounds-checked loads with negative register offsets, prefetches that cross clip boundaries,
    stretches that start on such frames)."""
    L = 12000
    x = np.zeros((B, L), np.float32)
    x[:, 0], x[:, 1], x[:, 2] = 1.0, -0.7, 0.5
    x[:, -1], x[:, -2], x[:, -3] = 0.9, -0.6, 0.4
    kw = dict(sr=16000, n_fft=n_fft, hop_length=hop, n_mels=n_mels)
    got = host(ap.melspectrogram(dev(x), **kw))
    want = ao.melspectrogram(x[:1], **kw)
    np.testing.assert_allclose(got, np.broadcast_to(want, got.shape), rtol=1e-4, atol=1e-7)
    S = host(ap.stft(dev(x), n_fft=n_fft, hop_length=hop))
    np.testing.assert_allclose(S, np.broadcast_to(ao.stft(x[0], n_fft=n_fft, hop_length=hop), S.shape), rtol=1e-4, atol=1e-6)
