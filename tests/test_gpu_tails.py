"""GPU parity for the tails of the hot path: resample_poly / resample, Griffin-Lim, dB
conversion, DCT and MFCC — HIP kernels through the C ABI vs the CPU oracle and the
committed SciPy fixtures.  Mirrors the reference's tests/test_resample.py,
test_griffinlim.py, test_convert.py and test_mfcc.py."""

import numpy as np
import pytest

from conftest import load_golden
from oracle import audio_oracle as ao

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import mlx_audio_primitives_amd as ap  # noqa: E402


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ resample
def test_resample_poly_matches_scipy_bit_for_bit():
    z = load_golden("resample_scipy.npz")
    for tag, up, down in (("p13", 1, 3), ("p21", 2, 1), ("p32", 3, 2), ("p147_160", 147, 160)):
        got = host(ap.resample_poly(dev(z[f"{tag}_y"]), up, down))
        assert got.shape == z[f"{tag}_out"].shape
        np.testing.assert_array_equal(got, z[f"{tag}_out"], err_msg=tag)


def test_resample_poly_api(batch_signals):
    y = dev(batch_signals)
    assert ap.resample_poly(y, 1, 2).shape == (4, 11025)                      # length rule
    assert ap.resample_poly(y, 3, 1).shape == (4, 66150)
    assert ap.resample_poly(y[0], 2, 4).shape == (11025,)                      # ratio reduced
    assert ap.resample_poly(y, 5, 5) is y                                      # identity early-out
    a = host(ap.resample_poly(y, 2, 4))
    np.testing.assert_array_equal(a, host(ap.resample_poly(y, 1, 2)))
    np.testing.assert_array_equal(a, ao.resample_poly(batch_signals, 1, 2))
    yt = dev(np.ascontiguousarray(batch_signals[:, :3000].T))                  # axis=0
    np.testing.assert_array_equal(host(ap.resample_poly(yt, 1, 3, axis=0)),
                                  ao.resample_poly(batch_signals[:, :3000].T, 1, 3, axis=0))
    with pytest.raises(ValueError, match="must be positive"):
        ap.resample_poly(y, 0, 1)


@pytest.mark.parametrize("down,L", [(2, 50001), (3, 48000), (4, 22050), (5, 9999), (7, 3000), (8, 70001), (3, 50)])
def test_resample_poly_decimator_bit_exact(down, L):
    """LDS-tiled register-blocked decimator (up == 1) against SciPy, bit for bit."""
    import scipy.signal
    rng = np.random.default_rng(down * 1000 + L)
    x = rng.standard_normal((3, L)).astype(np.float32)
    want = scipy.signal.resample_poly(x, 1, down, axis=-1).astype(np.float32)
    np.testing.assert_array_equal(host(ap.resample_poly(dev(x), 1, down)), want)


def test_resample_poly_48k_to_16k_config4_slice():
    g = torch.Generator(device="cuda").manual_seed(4)
    y = torch.randn((8, 480000), device="cuda", generator=g)
    out = ap.resample_poly(y, 16000, 48000)            # gcd-reduces to 1:3
    assert out.shape == (8, 160000)
    np.testing.assert_array_equal(host(out[3]), ao.resample_poly(host(y[3]), 1, 3))


def test_resample_linear_and_errors(random_signal):
    y = dev(random_signal)
    out = ap.resample(y, 22050, 16000, res_type="linear")
    np.testing.assert_allclose(host(out), ao.resample(random_signal, 22050, 16000, res_type="linear"),
                               rtol=1e-6, atol=1e-6)
    assert ap.resample(y, 22050, 22050) is y
    with pytest.raises(ValueError, match="Unknown res_type"):
        ap.resample(y, 22050, 16000, res_type="sinc")


def test_resample_fft_matches_scipy_fixtures_and_oracle(random_signal):
    # fixtures: scipy.signal.resample itself = the reference's implementation (resample.py:97,123)
    z = load_golden("resample_scipy.npz")
    for tag, num in (("f_half", 2400), ("f_up", 1500), ("f_odd", 367), ("f_b", 1102)):
        x = z[f"{tag}_y"]
        x2 = x if x.ndim == 2 else x[None]
        L = x2.shape[-1]
        got = host(ap.resample(dev(x2), L, num, res_type="fft"))     # ratio num/L -> target length num
        want = z[f"{tag}_out"] if x.ndim == 2 else z[f"{tag}_out"][None]
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4, err_msg=tag)
    # tests/test_resample.py:39-99: 2x down, 2x up, odd ratios, scale
    y = dev(random_signal)
    for orig, target in ((22050, 11025), (11025, 22050), (22050, 16000), (44100, 48000)):
        got = host(ap.resample(y, orig, target))
        want = ao.resample(random_signal, orig, target)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(ap.resample(y, 22050, 16000, scale=True)),
                               ao.resample(random_signal, 22050, 16000, scale=True), rtol=1e-4, atol=1e-4)
    assert ap.resample(y, 22050, 16000, fix=False).shape == ao.resample(random_signal, 22050, 16000, fix=False).shape


@pytest.mark.parametrize("L,num,B", [(9001, 4500, 2), (10007, 5003, 1), (100003, 36283, 1), (44100, 40009, 3),
                                     (2 * 8191, 8191, 2), (300007, 150001, 1)])
def test_resample_fft_lengths_with_large_prime_factors(L, num, B):
    """Lengths the four-step engine cannot factor (a prime factor > 4096) run as chirp-z convolutions on
    it; reference resample.py:97,123 = scipy.signal.resample for any length."""
    import scipy.signal
    x = np.random.default_rng(L).standard_normal((B, L)).astype(np.float32)
    got = host(ap.resample(dev(x), L, num, res_type="fft"))
    want = scipy.signal.resample(x.astype(np.float64), num, axis=-1)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(got, ao.resample(x, L, num, res_type="fft"), rtol=1e-4, atol=5e-5)


# ------------------------------------------------------------------ dB / DCT / MFCC
def test_db_conversions():
    rng = np.random.default_rng(5)
    S = (rng.standard_normal((4, 128, 44)).astype(np.float32)) ** 2
    np.testing.assert_allclose(host(ap.power_to_db(dev(S))), ao.power_to_db(S), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(host(ap.power_to_db(dev(S), top_db=None)), ao.power_to_db(S, top_db=None),
                               rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(host(ap.power_to_db(dev(S), ref=torch.max)), ao.power_to_db(S, ref=np.max),
                               rtol=1e-5, atol=1e-4)
    A = np.sqrt(S)
    np.testing.assert_allclose(host(ap.amplitude_to_db(dev(A))), ao.amplitude_to_db(A), rtol=1e-5, atol=1e-4)
    db = ao.power_to_db(S)
    np.testing.assert_allclose(host(ap.db_to_power(dev(db))), ao.db_to_power(db), rtol=1e-5)
    np.testing.assert_allclose(host(ap.db_to_amplitude(dev(db))), ao.db_to_amplitude(db), rtol=1e-5)
    with pytest.raises(ValueError, match="top_db must be positive"):
        ap.power_to_db(dev(S), top_db=-1.0)
    # the clip is against the GLOBAL max of the batched array (convert.py:58)
    S2 = np.stack([S[0], S[0] * 1e-12])
    out = host(ap.power_to_db(dev(S2)))
    assert out[1].min() == pytest.approx(out.max() - 80.0)


def test_dct_matches_scipy_fixture():
    z = load_golden("dct_scipy.npz")
    x = dev(z["x"])
    np.testing.assert_allclose(host(ap.dct(x)), z["ortho_full"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(ap.dct(x, n=13)), z["ortho_13"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(ap.dct(x, norm=None)), z["none_full_half_scipy"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(ap._ext.dct(x, 13, -1, "ortho")), z["ortho_13"], rtol=1e-4, atol=1e-4)
    xt = dev(np.ascontiguousarray(z["x"].T))
    np.testing.assert_allclose(host(ap.dct(xt, n=13, axis=0)), z["ortho_13"].T, rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError, match="Only DCT type 2"):
        ap.dct(x, type=3)


@pytest.mark.parametrize("n_mfcc,n_mels", [(13, 40), (20, 80), (40, 128)])
def test_mfcc(random_signal, n_mfcc, n_mels):
    got = host(ap.mfcc(dev(random_signal), sr=22050, n_mfcc=n_mfcc, n_mels=n_mels))
    want = ao.mfcc(random_signal, sr=22050, n_mfcc=n_mfcc, n_mels=n_mels)
    assert got.shape == (n_mfcc, 44)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-3)     # dB of tiny bins amplifies 1e-7 rel errors


def test_mfcc_variants(batch_signals):
    y = batch_signals[:, :8000]
    got = host(ap.mfcc(dev(y), sr=16000, n_mfcc=13, n_fft=400, hop_length=160, n_mels=40, lifter=22))
    want = ao.mfcc(y, sr=16000, n_mfcc=13, n_fft=400, hop_length=160, n_mels=40, lifter=22)
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=2e-3)
    S = ao.power_to_db(ao.melspectrogram(y[0], sr=16000, n_fft=400, hop_length=160, n_mels=40))
    np.testing.assert_allclose(host(ap.mfcc(S=dev(S), n_mfcc=13)), ao.mfcc(S=S, n_mfcc=13), rtol=1e-4, atol=1e-3)
    with pytest.raises(ValueError, match="must be positive"):
        ap.mfcc(dev(y), n_mfcc=0)


# ------------------------------------------------------------------ Griffin-Lim
def test_griffinlim_matches_oracle_one_and_few_iterations(chirp_signal):
    y = chirp_signal[:8192]
    S = ao.magnitude(ao.stft(y, n_fft=512, hop_length=128))
    for n_iter in (1, 4):
        got = host(ap.griffinlim(dev(S), n_iter=n_iter, hop_length=128, random_state=42, length=len(y)))
        want = ao.griffinlim(S, n_iter=n_iter, hop_length=128, random_state=42, length=len(y))
        # the iteration is chaotic in the long run; after a few steps float32 paths still agree
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=2e-3)


def test_griffinlim_reference_thresholds(chirp_signal):
    # tests/test_griffinlim.py:99-121 — MSE of |stft(y_hat)| vs S on the chirp: {16:10, 32:5, 64:2}
    y = dev(chirp_signal)
    S = ap.magnitude(ap.stft(y, n_fft=2048, hop_length=512))
    for n_iter, bound in ((16, 10.0), (32, 5.0), (64, 2.0)):
        yr = ap.griffinlim(S, n_iter=n_iter, random_state=42, length=len(chirp_signal))
        Sr = ap.magnitude(ap.stft(yr, n_fft=2048, hop_length=512))
        mse = float(((S - Sr) ** 2).mean())
        assert mse < bound, (n_iter, mse)
        assert yr.shape == (len(chirp_signal),)
    a = ap.griffinlim(S, n_iter=8, random_state=7)
    b = ap.griffinlim(S, n_iter=8, random_state=7)
    np.testing.assert_allclose(host(a), host(b), atol=1e-5)        # same seed -> same output
    Sb = torch.stack([S, S * 0.5])
    assert ap.griffinlim(Sb, n_iter=2, random_state=0).shape[0] == 2
    with pytest.raises(ValueError, match="Unknown init"):
        ap.griffinlim(S, init="bogus")
    with pytest.raises(ValueError, match="must be positive"):
        ap.griffinlim(S, n_iter=0)
    with pytest.raises(ValueError, match="momentum must be <"):
        ap.griffinlim(S, momentum=1.0)


def test_device_pcg64_phase_matches_numpy():
    from mlx_audio_primitives_amd.griffinlim import _random_phase
    for seed, shape in ((42, (3, 5, 7)), (7, (2, 1025, 216)), (0, (1, 1, 1))):
        want = np.random.default_rng(seed).uniform(-np.pi, np.pi, shape).astype(np.float32)
        np.testing.assert_array_equal(host(_random_phase(seed, shape, "cuda")), want)


def test_griffinlim_config3_slice():
    """cfg3 shape per clip (5 s @ 22.05 kHz, n_fft=2048 hop=512, 32 iterations), small batch."""
    g = torch.Generator(device="cuda").manual_seed(3)
    y = torch.randn((4, 110250), device="cuda", generator=g)
    S = ap.magnitude(ap.stft(y))
    assert S.shape == (4, 1025, 216)
    yr = ap.griffinlim(S, n_iter=32, momentum=0.99, init="random", random_state=42, length=110250)
    assert yr.shape == (4, 110250) and bool(torch.isfinite(yr).all())
    Sr = ap.magnitude(ap.stft(yr))
    rel = float(((S - Sr) ** 2).sum().sqrt() / (S ** 2).sum().sqrt())
    assert rel < 0.35, rel      # spectral convergence on white noise after 32 iterations
