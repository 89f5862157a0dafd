"""CPU: the C-ABI library builds, loads, exports every symbol include/audioprims.h
declares, and its host-side builders agree with the oracle / golden fixtures.
No device compute here."""

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import audio_oracle as ao

import mlx_audio_primitives_amd as ap
from mlx_audio_primitives_amd import _extension as ext


def test_library_loaded_and_exports_header_symbols():
    assert ap.HAS_HIP_EXT and ap._ext is not None
    header = open(os.path.join(ROOT, "include", "audioprims.h")).read()
    declared = set(re.findall(r"\b(ap_[a-z0-9_]+)\s*\(", header))
    assert declared == set(ext.ABI_SYMBOLS)
    lib = ext.lib()
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.ap_version() >= 100


def test_no_gpu_is_a_loud_error():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        ap.stft(np.zeros(4096, np.float32))
    with pytest.raises(RuntimeError, match="no HIP device"):
        ap.melspectrogram(np.zeros(4096, np.float32))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mlx-audio-primitives_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", ""), (dirpath, f)


def test_host_windows_match_scipy_and_oracle():
    z = load_golden("windows_scipy.npz")
    for key in z.files:
        name, n, periodic = key.rsplit("_", 2)
        w = ext.generate_window_host(name, int(n), bool(int(periodic)))
        np.testing.assert_allclose(w, z[key], rtol=1e-5, atol=1e-5, err_msg=key)
        np.testing.assert_array_equal(w, ao.get_window(name, int(n), bool(int(periodic)), native=True))
    for name in ("hann", "hamming", "blackman", "bartlett"):
        for n in (16, 255, 2048):
            w = ext.generate_window_host(name, n, False)
            assert np.array_equal(w, w[::-1])
    with pytest.raises(ValueError, match="Unknown window type"):
        ap.get_window("kaiser", 16)
    with pytest.raises(ValueError, match="must match n_fft"):
        ap.get_window(np.ones(8, np.float32), 16)
    with pytest.raises(TypeError):
        ap.get_window(3, 16)
    # periodic == symmetric(n+1)[:-1]  (tests/test_mathematical_properties.py:663-688)
    np.testing.assert_array_equal(ext.generate_window_host("hann", 64, True),
                                  ext.generate_window_host("hann", 65, False)[:64])


def test_host_mel_filterbank():
    z = load_golden("mel_filters.npz")
    for row in z["meta"]:
        name, sr, n_fft, n_mels, fmin, fmax, norm, scale = str(row).split(",")
        norm = None if norm == "None" else norm
        htk = scale == "htk"
        args = (int(sr), int(n_fft), int(n_mels), float(fmin), float(fmax))
        fb_py = ap.mel_filterbank(*args, htk=htk, norm=norm, device="cpu").numpy()
        np.testing.assert_allclose(fb_py, z[name], rtol=1e-5, atol=1e-5, err_msg=name)
        np.testing.assert_array_equal(fb_py, ao.mel_filterbank(*args, htk=htk, norm=norm))
        fb_c = ext.mel_filterbank_host(*args, htk=htk, norm=norm or "")
        np.testing.assert_allclose(fb_c, z[name], rtol=1e-5, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(fb_c, ao.mel_filterbank_native(*args, htk=htk, norm=norm),
                                   rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError, match="cannot exceed Nyquist"):
        ap.mel_filterbank(22050, 2048, 128, fmax=12000.0)
    with pytest.raises(ValueError, match="must be positive"):
        ap.mel_filterbank(22050, 2048, 0)
    with pytest.raises(ValueError, match="must be less than fmax"):
        ap.mel_filterbank(22050, 2048, 128, fmin=5000.0, fmax=4000.0)
    with pytest.raises(ValueError, match="Nyquist"):
        ext.mel_filterbank_host(22050, 2048, 128, 0.0, 12000.0)


def test_host_mel_scale():
    z = load_golden("mel_filters.npz")
    np.testing.assert_allclose(ap.hz_to_mel(z["htk_hz"], htk=True), z["htk_mel"], rtol=1e-12)
    np.testing.assert_allclose(ap.hz_to_mel(z["slaney_hz"]), z["slaney_mel"], atol=1e-12)
    np.testing.assert_allclose(ap.mel_to_hz(z["slaney_mel"]), z["slaney_hz"], atol=1e-9)
    np.testing.assert_allclose(ap._ext.hz_to_mel(z["htk_hz"], htk=True), z["htk_mel"], rtol=1e-6)
    np.testing.assert_allclose(ap._ext.mel_to_hz(z["slaney_mel"]), z["slaney_hz"], rtol=1e-6, atol=1e-4)


def test_host_dct_matrix_and_twiddles():
    C = ext.dct_matrix_host(13, 128, "ortho")
    np.testing.assert_allclose(C, ao.dct_matrix(13, 128, "ortho"), atol=1e-5)
    z = load_golden("dct_scipy.npz")
    np.testing.assert_allclose(z["x"].astype(np.float64) @ C.T.astype(np.float64), z["ortho_13"],
                               rtol=1e-4, atol=1e-4)
    for n in (8, 400, 2048, 27):
        tw = ext.twiddle_table_host(n).reshape(n, 2).astype(np.float64)
        a = 2 * np.pi * np.arange(n) / n
        np.testing.assert_allclose(tw[:, 0], np.cos(a), atol=6e-8)
        np.testing.assert_allclose(tw[:, 1], np.sin(a), atol=6e-8)
    tw = ext.twiddle_table_host(8).reshape(8, 2)
    assert tw[2, 0] == 0.0 and tw[4, 1] == 0.0 and tw[6, 0] == 0.0
    assert ext.lib().ap_fft_supported(2048) == 1 and ext.lib().ap_fft_supported(400) == 1


def test_check_nola_and_validation_messages():
    assert ap.check_nola("hann", 512, 2048)
    assert not ap.check_nola("hann", 2048, 2048)
    with pytest.raises(ValueError, match="must be positive"):
        ap.validate_positive(0, "n_iter")
    with pytest.raises(ValueError, match="must be < 1.0"):
        ap.validate_range(1.0, "momentum", min_val=0.0, max_val=1.0, max_inclusive=False)


@pytest.mark.parametrize("window", ["hann", "hamming", "blackman", "bartlett", "rectangular"])
def test_check_nola_matches_oracle_on_a_grid(window):
    """stft.py:382-431 (scipy.signal.check_NOLA): hops that divide n_fft, ragged hops, hop = n_fft
    (hann / bartlett / blackman start at 0 -> False), hop = 1, and array windows."""
    for n_fft in (16, 64, 400, 512, 2048):
        for hop in (1, 3, n_fft // 4, n_fft // 3, n_fft // 2, n_fft // 2 + 1, n_fft - 1, n_fft):
            assert ap.check_nola(window, hop, n_fft) == ao.check_nola(window, hop, n_fft), (n_fft, hop)
    w = ao.get_window(window, 64)
    w[3::8] = 0.0                                   # position 3 of every 8-sample hop is never covered
    for hop in (8, 16, 12, 64):
        assert ap.check_nola(w, hop, 64) == ao.check_nola(w, hop, 64)
    assert ap.check_nola(w, 8, 64) is False and ap.check_nola(w, 12, 64) is True
    assert ap.check_nola(window, 16, 64, tol=1e3) == ao.check_nola(window, 16, 64, tol=1e3) is False


def test_cpu_baseline_variant_matches_oracle():
    y = ao.random_signal(8000)
    a = ao.melspectrogram_cpu_baseline(y[None], workers=2)
    b = ao.melspectrogram(y[None])
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-4)


def test_host_resample_poly_taps_are_scipys_bit_for_bit():
    """ap_resample_poly_taps_host == the float32 filter scipy.signal.resample_poly builds (firwin + Kaiser
    5.0, cast, * up, leading zeros) in every bit - including the 1e-17-sized values numpy's sinc leaves
    at the filter's zero crossings, which decide the last bits of outputs that nearly cancel."""
    from scipy.signal import firwin
    from mlx_audio_primitives_amd.resample import _poly_taps_host
    for up, down in ((3, 2), (1, 3), (160, 147), (2, 1), (1, 2), (1, 8), (5, 7), (147, 160), (7, 1), (1, 5)):
        raw, n_pre_remove = _poly_taps_host(up, down)
        taps = np.frombuffer(raw, dtype=np.float32)
        max_rate = max(up, down)
        half_len = 10 * max_rate
        h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(np.float32)
        h *= np.float32(up)
        n_pre_pad = down - half_len % down
        np.testing.assert_array_equal(taps, np.concatenate([np.zeros(n_pre_pad, np.float32), h]))
        assert n_pre_remove == (half_len + n_pre_pad) // down


def test_fused_predicates_agree_with_the_launch_bounds():
    """ap_spectral_audio_fused / ap_melspec_pcm16_fused must say "no" for clips beyond the bounds the
    fused kernels' launch code enforces (L <= 2^28 samples, T <= 2^24 frames), so that the Python layer takes the
    two-kernel / scratch route instead of failing inside the launch.  Host-only: the predicates touch no GPU."""
    import ctypes

    import numpy as np

    from mlx_audio_primitives_amd import _extension as ext

    L_ = ext.lib()
    desc = np.zeros(16, np.int32)
    desc[0] = 2 | 1            # AP_PLAN_PARTS | AP_PLAN_BANDED (include/audioprims.h)
    desc[12], desc[15] = 128, 3
    dptr = desc.ctypes.data_as(ctypes.c_void_p)
    L_.ap_spectral_audio_fused.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L_.ap_melspec_pcm16_fused.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    assert L_.ap_spectral_audio_fused(220500, 2048, 512, 1, 0) == 1
    assert L_.ap_spectral_audio_fused((1 << 28) + 2, 2048, 512, 1, 0) == 0          # sample offsets past 32 bits
    assert L_.ap_spectral_audio_fused(1 << 28, 2048, 2, 1, 0) == 0                  # more than 2^24 frames
    ok = L_.ap_melspec_pcm16_fused(220500, 2048, 512, 1, 0, 128, 2.0, dptr)
    if ok:                       # (the plan flags above are what mel_plan_host sets for the default bank)
        assert L_.ap_melspec_pcm16_fused((1 << 28) + 2, 2048, 512, 1, 0, 128, 2.0, dptr) == 0
        assert L_.ap_melspec_pcm16_fused(1 << 28, 2048, 2, 1, 0, 128, 2.0, dptr) == 0
