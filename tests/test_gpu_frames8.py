"""GPU parity for the eight-frames-per-wave kernels (csrc/kernels_frames8.h: n_fft 256 / 400 / 512,
STFT and mel-spectrogram) against the CPU oracle: the reference's own n_fft = 512 grid
(/root/reference/tests/test_stft.py:45-59), more groups than workgroups, odd hops (sample pairs that
straddle the clip ends), hop == n_fft, ragged last groups, wide bands (64-bin weight rows), and the
shapes that fall back to the LDS engine (reflect padding, > 128 filters, bands over 64 bins)."""

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


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "gpu-marked tests need an MI355X"
    assert ap.HAS_HIP_EXT
    yield
    torch.cuda.synchronize()


@pytest.mark.parametrize("n_fft,hop,B,L,center", [
    (512, 128, 3, 22050, True), (512, 256, 2, 22051, True), (512, 512, 2, 10000, False),
    (512, 77, 1, 5001, True), (400, 160, 4, 16000, True), (400, 33, 1, 4000, False),
    (256, 64, 5, 8000, True), (256, 255, 2, 3001, True),
    (512, 128, 300, 9000, True),            # 2 700 groups on 256 workgroups: stretches cross clip ends
    (400, 160, 700, 4000, False),
])
def test_stft_frames8(n_fft, hop, B, L, center):
    rng = np.random.default_rng(n_fft + hop + B)
    y = rng.standard_normal((B, L)).astype(np.float32)
    S = host(ap.stft(dev(y), n_fft=n_fft, hop_length=hop, center=center))
    idx = sorted(set([0, B - 1, B // 2]))
    for b in idx:
        want = ao.stft(y[b], n_fft=n_fft, hop_length=hop, center=center)
        assert S[b].shape == want.shape
        np.testing.assert_allclose(S[b], want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n_fft,kw,B,L", [
    (512, dict(sr=22050, hop_length=128, n_mels=128), 3, 22050),
    (512, dict(sr=22050, hop_length=256, n_mels=40, power=1.0), 2, 12001),            # bands of 39 bins: 64-float rows
    (512, dict(sr=16000, hop_length=160, n_mels=80, center=False, htk=True), 2, 16000),
    (512, dict(sr=22050, hop_length=128, n_mels=64, power=1.5), 300, 9000),           # more groups than workgroups
    (256, dict(sr=8000, hop_length=64, n_mels=32), 4, 8000),
    (256, dict(sr=16000, hop_length=100, n_mels=64, fmin=100.0, norm=None), 2, 5000),
    (400, dict(sr=16000, hop_length=160, n_mels=40), 2, 16000),                       # Whisper kernel, wide rows
    # index-remapped edge frames (PADGEN instantiation), then the fallbacks to the LDS engine
    (512, dict(sr=22050, hop_length=128, n_mels=64, pad_mode="reflect"), 2, 9000),
    (400, dict(sr=16000, hop_length=160, n_mels=80, pad_mode="edge"), 300, 4000),
    (512, dict(sr=22050, hop_length=128, n_mels=160), 2, 9000),
    (512, dict(sr=22050, hop_length=128, n_mels=10), 2, 9000),
])
def test_melspectrogram_frames8(n_fft, kw, B, L):
    rng = np.random.default_rng(n_fft + B + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    S = host(ap.melspectrogram(dev(y), n_fft=n_fft, **kw))
    idx = sorted(set([0, B - 1, B // 2]))
    want = ao.melspectrogram(y[idx], n_fft=n_fft, **kw)
    assert S[idx].shape == want.shape
    np.testing.assert_allclose(S[idx], want, rtol=1e-4, atol=1e-4 * max(1.0, float(want.max()) * 1e-2))


@pytest.mark.parametrize("n_fft,hop,L", [(2048, 511, 30001), (2048, 333, 22051), (512, 77, 5001), (400, 33, 4001)])
def test_melspectrogram_odd_hop_and_length(n_fft, hop, L):
    """Odd hops put odd sample indices at the head of a lane's 8-byte loads; those shapes take the
    kernels with 4-byte loads.  Odd clip lengths leave a pair straddling the clip end."""
    rng = np.random.default_rng(hop)
    y = rng.standard_normal((3, L)).astype(np.float32)
    kw = dict(sr=22050, n_fft=n_fft, hop_length=hop, n_mels=64)
    np.testing.assert_allclose(host(ap.melspectrogram(dev(y), **kw)), ao.melspectrogram(y, **kw), rtol=1e-4, atol=1e-3)
    kw = dict(sr=22050, n_fft=n_fft, hop_length=hop + 1, n_mels=64)
    np.testing.assert_allclose(host(ap.melspectrogram(dev(y), **kw)), ao.melspectrogram(y, **kw), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("n_fft,hop", [(2048, 511), (2048, 1), (1024, 255), (512, 77), (400, 33), (256, 3), (64, 5)])
def test_stft_odd_hop_every_engine(n_fft, hop):
    """Centred frames at odd sample offsets: sample -1 and sample 0 share a lane's load."""
    rng = np.random.default_rng(n_fft + hop)
    L = 4 * n_fft + 1 if hop > 1 else n_fft + 7
    y = rng.standard_normal((2, L)).astype(np.float32)
    S = host(ap.stft(dev(y), n_fft=n_fft, hop_length=hop))
    for b in range(2):
        np.testing.assert_allclose(S[b], ao.stft(y[b], n_fft=n_fft, hop_length=hop), rtol=1e-4, atol=1e-4)


def test_frames8_mfcc_global_max_and_shuffle():
    """The fused global maximum of the n_fft = 512 kernel feeds mfcc's top_db clip; clip order is
    irrelevant to every clip's frames (bitwise)."""
    rng = np.random.default_rng(5)
    y = (rng.standard_normal((6, 12000)) * np.array([1.0, 1e-3, 0.1, 1.0, 1e-4, 0.5])[:, None]).astype(np.float32)
    kw = dict(sr=22050, n_mfcc=13, n_fft=512, hop_length=128, n_mels=64)
    got = host(ap.mfcc(dev(y), **kw))
    np.testing.assert_allclose(got, ao.mfcc(y, **kw), rtol=1e-3, atol=2e-3)
    perm = np.array([3, 0, 5, 1, 4, 2])
    S = host(ap.melspectrogram(dev(y), sr=22050, n_fft=512, hop_length=128, n_mels=64))
    Sp = host(ap.melspectrogram(dev(y[perm]), sr=22050, n_fft=512, hop_length=128, n_mels=64))
    np.testing.assert_array_equal(Sp, S[perm])


@pytest.mark.parametrize("n_fft,hop,B,L", [(512, 128, 3, 22050), (512, 256, 2, 9001), (512, 64, 2, 5000), (400, 160, 4, 16000),
                                           (400, 100, 2, 4100), (256, 64, 5, 8000), (256, 33, 1, 1500),
                                           (512, 128, 300, 9000), (400, 160, 700, 4000)])
def test_istft_frames8_fused(n_fft, hop, B, L):
    """Fused ISTFT of the frames8 family (irfft + overlap-add + normalisation + trim in one kernel) against
    the oracle and the two-kernel route (AP_ISTFT8_UNFUSED): round trip <= 1e-5, `length` shorter / longer,
    stretches that start inside clips and cross clip ends."""
    import os
    rng = np.random.default_rng(n_fft + hop + B)
    y = rng.standard_normal((B, L)).astype(np.float32)
    S = ap.stft(dev(y), n_fft=n_fft, hop_length=hop)
    yr = host(ap.istft(S, hop_length=hop, length=L))
    np.testing.assert_allclose(yr, y, atol=1e-5)                          # README.md:118
    Sh = host(S)
    for b in sorted(set([0, B - 1])):
        for length in (L - 333, L + 40):
            got = host(ap.istft(S[b], hop_length=hop, length=length))
            want = ao.istft(Sh[b], hop_length=hop, n_fft=n_fft, length=length)
            np.testing.assert_allclose(got[:L + 40], want[:L + 40], atol=1e-5)
    os.environ["AP_ISTFT8_UNFUSED"] = "1"
    try:
        y2 = host(ap.istft(S, hop_length=hop, length=L))
    finally:
        del os.environ["AP_ISTFT8_UNFUSED"]
    np.testing.assert_allclose(yr, y2, atol=2e-6)


def test_istft_frames8_no_centre_and_fallback_hops():
    rng = np.random.default_rng(2)
    y = rng.standard_normal((2, 6000)).astype(np.float32)
    S = ap.stft(dev(y), n_fft=512, hop_length=128, center=False)
    got = host(ap.istft(S, hop_length=128, center=False))
    want = ao.istft(host(S), hop_length=128, n_fft=512, center=False)
    assert got.shape == want.shape
    np.testing.assert_allclose(got[:, 512:-512], want[:, 512:-512], atol=1e-5)
    # hop below n_fft / 8: the carry of one group no longer covers the overlap - two-kernel route
    S = ap.stft(dev(y), n_fft=512, hop_length=32)
    np.testing.assert_allclose(host(ap.istft(S, hop_length=32, length=6000)), y, atol=1e-5)


@pytest.mark.parametrize("pad_mode", ["reflect", "edge"])
@pytest.mark.parametrize("n_fft,hop,B,L", [(512, 128, 3, 9001), (400, 160, 200, 4000), (256, 64, 2, 3000), (2048, 512, 3, 30001),
                                           (1024, 256, 3, 20001)])
def test_stft_and_mel_pad_modes_on_the_wave_kernels(pad_mode, n_fft, hop, B, L):
    """reflect / edge padding on the eight-frames-per-wave kernels and the n_fft = 2048 run kernel: only the
    frames that reach over a clip end take the index remap."""
    rng = np.random.default_rng(n_fft + B)
    y = rng.standard_normal((B, L)).astype(np.float32)
    S = host(ap.stft(dev(y), n_fft=n_fft, hop_length=hop, pad_mode=pad_mode))
    M = host(ap.melspectrogram(dev(y), sr=16000, n_fft=n_fft, hop_length=hop, n_mels=40, pad_mode=pad_mode))
    for b in sorted(set([0, B - 1])):
        np.testing.assert_allclose(S[b], ao.stft(y[b], n_fft=n_fft, hop_length=hop, pad_mode=pad_mode), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(M[b], ao.melspectrogram(y[b], sr=16000, n_fft=n_fft, hop_length=hop, n_mels=40,
                                                           pad_mode=pad_mode), rtol=1e-4, atol=1e-3)
