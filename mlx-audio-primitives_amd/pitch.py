"""Autocorrelation, pitch_detect_acf, periodicity — same API as /root/reference/mlx_audio_primitives/pitch.py
(native mirror csrc/primitives/autocorrelation.cpp:10-84).

Wiener-Khinchin on the device: centre, zero-pad to the next power of two >= 2 n - 1, forward and
inverse four-step FFT (both legs LDS-resident, the transform ``resample(res_type="fft")`` uses),
|.|^2 in between, first max_lag lags, normalised by r[0].  The reference's Python path does the two
FFTs in NumPy on the host (pitch.py:88-99).
"""

from __future__ import annotations

import ctypes

import torch

from . import _extension as _x
from .stft import _get_twiddles


def autocorrelation(y, max_lag: int | None = None, normalize: bool = True, center: bool = True) -> torch.Tensor:
    """r[k] = sum_n y[n] y[n + k] for k < max_lag (default: all n lags), (max_lag,) or (batch, max_lag)."""
    y = _x.to_device_f32(y)
    if y.ndim not in (1, 2):
        raise ValueError("signal must be 1-dimensional (samples,) or 2-dimensional (batch, samples)")
    one_d = y.ndim == 1
    if one_d:
        y = y[None, :]
    y = y.contiguous()
    B, n = y.shape
    if max_lag is None or max_lag <= 0:          # bindings.cpp: -1 = all lags
        max_lag = n
    max_lag = min(int(max_lag), n)
    dev = y.device
    out = torch.empty((B, max_lag), dtype=torch.float32, device=dev)
    if B > 0 and n > 0:
        lib = _x.lib()
        N = int(lib.ap_autocorrelation_nfft(n))
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        if lib.ap_cfft_split_host(N, ctypes.addressof(a), ctypes.addressof(b)) != 0:
            raise ValueError(f"autocorrelation: signals of {n} samples need a {N}-point transform, beyond "
                             "the on-chip four-step FFT (4096 x 4096)")
        tw1, tw2 = _get_twiddles(a.value, dev), _get_twiddles(b.value, dev)
        ws = torch.empty(4 * B * N + B, dtype=torch.float32, device=dev)
        _x.check(_x.dlib(dev).ap_autocorrelation_f32(_x.ptr(y), B, n, max_lag, int(bool(normalize)),
                                                     int(bool(center)), _x.ptr(tw1), _x.ptr(tw2), _x.ptr(ws),
                                                     _x.ptr(out), _x.stream_ptr(dev)))
    return out[0] if one_d else out


def _frame_acf_peaks(y, sr, fmin, fmax, frame_length, hop_length, threshold, center, want):
    """Shared body of pitch_detect_acf / periodicity (reference pitch.py:118-369): constant centre padding,
    frames, the raw autocorrelation of every centred frame on the device (the four-step FFT pair of
    `autocorrelation`), then one pass that picks the peak of r / r[0] in the lag range.  The reference walks
    the frames in a Python loop with two NumPy FFTs each."""
    from ._validation import validate_positive

    validate_positive(frame_length, "frame_length")
    validate_positive(hop_length, "hop_length")
    if want == "pitch" and fmin >= fmax:
        raise ValueError(f"fmin ({fmin}) must be less than fmax ({fmax})")
    min_lag, max_lag = int(sr / fmax), int(sr / fmin)
    y = _x.to_device_f32(y)
    one_d = y.ndim == 1
    if one_d:
        y = y[None, :]
    if y.ndim != 2:
        raise ValueError(f"y must be 1D or 2D, got {y.ndim}D")
    y = y.contiguous()
    B, L = y.shape
    dev = y.device
    pad = frame_length // 2 if center else 0
    Lp = L + 2 * pad
    T = 1 + (Lp - frame_length) // hop_length if Lp >= frame_length else 0
    f0 = torch.zeros((B, max(T, 0)), dtype=torch.float32, device=dev)
    voiced = torch.zeros((B, max(T, 0)), dtype=torch.uint8, device=dev)
    per = torch.zeros((B, max(T, 0)), dtype=torch.float32, device=dev)
    if B > 0 and T > 0:
        d = _x.dlib(dev)
        st = _x.stream_ptr(dev)
        yp = y
        if pad:
            yp = torch.empty((B, Lp), dtype=torch.float32, device=dev)
            _x.check(d.ap_pad_f32(_x.ptr(y), B, L, pad, _x.PAD_MODES["constant"], _x.ptr(yp), st))
        # Lags beyond the frame are treated as 0 here.  Deviation from the reference when sr / fmin >= frame_length
        # (e.g. frame_length = 1024, sr = 22050, fmin = 20): its r is the full n_fft-long irfft, so it then sees the
        # mirrored negative lags in r[frame_length:] and a slice cut at n_fft (pitch.py:189-214).  No fixture of the
        # reference covers that shape ("parity unpinned"); the search range inside the frame is identical.
        n_lag = min(max_lag + 1, frame_length)
        # one clip's frames at a time (a few clips per pass when they are short): bounded workspace
        per_pass = max(1, min(B, (1 << 15) // max(T, 1)))
        for b0 in range(0, B, per_pass):
            nb = min(per_pass, B - b0)
            frames = torch.empty((nb * T, frame_length), dtype=torch.float32, device=dev)
            _x.check(d.ap_frame_f32(_x.ptr(yp[b0:b0 + nb]), nb, Lp, int(frame_length), int(hop_length), _x.ptr(frames), st))
            r = autocorrelation(frames, max_lag=n_lag, normalize=False, center=True)
            _x.check(d.ap_acf_peaks_f32(_x.ptr(r), nb * T, n_lag, min_lag, max_lag, float(threshold), float(sr),
                                        _x.ptr(f0[b0:b0 + nb]), _x.ptr(voiced[b0:b0 + nb]), _x.ptr(per[b0:b0 + nb]), st))
    return f0, voiced.bool(), per, one_d


def pitch_detect_acf(y, sr: int = 22050, fmin: float = 50.0, fmax: float = 2000.0, frame_length: int = 2048,
                     hop_length: int = 512, threshold: float = 0.1, center: bool = True):
    """(f0, voiced_flag) per frame: f0 = sr / lag of the first local maximum above `threshold` of the normalised
    autocorrelation in [sr / fmax, sr / fmin] (reference pitch.py:118-264).  (n_frames,) or (batch, n_frames)."""
    f0, voiced, _, one_d = _frame_acf_peaks(y, sr, fmin, fmax, frame_length, hop_length, threshold, center, "pitch")
    return (f0[0], voiced[0]) if one_d else (f0, voiced)


def periodicity(y, sr: int = 22050, fmin: float = 50.0, fmax: float = 2000.0, frame_length: int = 2048,
                hop_length: int = 512, center: bool = True) -> torch.Tensor:
    """Largest normalised autocorrelation in the lag range per frame, (1, n_frames) or (batch, 1, n_frames)
    (reference pitch.py:267-369)."""
    _, _, per, one_d = _frame_acf_peaks(y, sr, fmin, fmax, frame_length, hop_length, 0.0, center, "periodicity")
    per = per[:, None, :]
    return per[0] if one_d else per
