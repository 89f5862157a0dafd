"""Mel scale, mel filterbank and mel spectrogram — same API as
/root/reference/mlx_audio_primitives/mel.py.

``hz_to_mel`` / ``mel_to_hz`` / the filterbank are host NumPy float64 exactly as in
the reference (mel.py:31-168; the reference keeps these off the device on
purpose).  ``melspectrogram`` is ONE fused kernel launch: the reference's
stft -> abs -> power -> matmul chain (mel.py:310-350) with no (B,F,T) intermediate.
"""

from __future__ import annotations

from functools import lru_cache

import numpy as np
import torch

from . import _extension as _x
from ._validation import validate_non_negative, validate_positive
from .stft import _frame_count, _get_padded_window, _get_twiddles, _resolve_stft_args
from .windows import _default_device


class _MelScale:
    """The two mel scales the reference offers (mel.py:24-93), float64 on the host.

    Slaney: linear below the 1 kHz knee (200/3 Hz per mel), logarithmic above it with
    27 mels per factor 6.4; HTK: 2595 log10(1 + f/700)."""

    knee_hz = 1000.0
    hz_per_mel = 200.0 / 3
    knee_mel = knee_hz / hz_per_mel
    log_step = np.log(6.4) / 27.0

    @classmethod
    def forward(cls, hz: np.ndarray, htk: bool) -> np.ndarray:
        if htk:
            return 2595.0 * np.log10(1.0 + hz / 700.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            above = cls.knee_mel + np.log(hz / cls.knee_hz) / cls.log_step
        return np.where(hz < cls.knee_hz, (hz - 0.0) / cls.hz_per_mel, above)

    @classmethod
    def inverse(cls, mel: np.ndarray, htk: bool) -> np.ndarray:
        if htk:
            return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
        above = cls.knee_hz * np.exp(cls.log_step * (mel - cls.knee_mel))
        return np.where(mel < cls.knee_mel, 0.0 + cls.hz_per_mel * mel, above)


def hz_to_mel(frequencies, htk: bool = False) -> np.ndarray:
    """Hz -> mel on the host (reference mel.py:31-62)."""
    return _MelScale.forward(np.asarray(frequencies), htk)


def mel_to_hz(mels, htk: bool = False) -> np.ndarray:
    """mel -> Hz on the host (reference mel.py:65-93)."""
    return _MelScale.inverse(np.asarray(mels), htk)


@lru_cache(maxsize=64)
def _triangular_bank(sr, n_fft, n_mels, fmin, fmax, htk, norm) -> np.ndarray:
    """Dense (n_mels, n_fft//2+1) float32 bank of triangles whose corners are equally spaced on
    the mel scale.  Bit-compatible with the reference's NumPy builder (mel.py:100-168), whose
    two quirks are part of the contract: each ramp's denominator carries a +1e-10 guard, and the
    triangle is rounded to float32 BEFORE the Slaney area normalisation 2 / (f[i+2] - f[i])
    (a float64 product rounded to float32 once more)."""
    if norm not in ("slaney", None):
        raise ValueError(f"Unknown norm: '{norm}'. Supported: 'slaney', None")
    corners = mel_to_hz(np.linspace(hz_to_mel(fmin, htk=htk), hz_to_mel(fmax, htk=htk), n_mels + 2),
                        htk=htk)
    left, peak, right = corners[:-2], corners[1:-1], corners[2:]
    grid = np.linspace(0, sr / 2.0, 1 + n_fft // 2)[np.newaxis, :]
    rising = (grid - left[:, np.newaxis]) / (peak - left + 1e-10)[:, np.newaxis]
    falling = (right[:, np.newaxis] - grid) / (right - peak + 1e-10)[:, np.newaxis]
    bank = np.maximum(0, np.minimum(rising, falling)).astype(np.float32)
    if norm == "slaney":
        np.multiply(bank, (2.0 / (right - left))[:, np.newaxis], out=bank, casting="same_kind")
    bank.setflags(write=False)
    return bank


_device_filterbank_cache: dict[tuple, tuple] = {}


def _mel_filterbank_full(sr, n_fft, n_mels, fmin, fmax, htk, norm, device):
    """(filterbank, contraction plan blob on `device`, host plan descriptor), cached."""
    validate_positive(n_mels, "n_mels")
    validate_non_negative(fmin, "fmin")
    if fmax is None:
        fmax = sr / 2.0
    if fmin >= fmax:
        raise ValueError(f"fmin ({fmin}) must be less than fmax ({fmax})")
    if fmax > sr / 2.0:
        raise ValueError(f"fmax ({fmax}) cannot exceed Nyquist frequency ({sr / 2.0})")
    dev = _default_device(device)
    key = (sr, n_fft, n_mels, fmin, fmax, htk, norm, str(dev))
    hit = _device_filterbank_cache.get(key)
    if hit is not None:
        return hit
    fb_np = _triangular_bank(sr, n_fft, n_mels, fmin, fmax, htk, norm)
    plan_np, desc = _x.mel_plan_host(fb_np)
    fb = torch.from_numpy(fb_np.copy()).to(dev)
    plan = torch.from_numpy(plan_np).to(dev)
    _device_filterbank_cache[key] = (fb, plan, desc)
    return fb, plan, desc


def mel_filterbank(sr: int, n_fft: int, n_mels: int = 128, fmin: float = 0.0,
                   fmax: float | None = None, htk: bool = False, norm: str | None = "slaney",
                   device=None) -> torch.Tensor:
    """Mel filterbank (n_mels, n_fft//2+1), float32, cached (reference mel.py:171-242)."""
    return _mel_filterbank_full(sr, n_fft, n_mels, fmin, fmax, htk, norm, device)[0]


def melspectrogram(y, sr: int = 22050, n_fft: int = 2048, hop_length: int | None = None,
                   win_length: int | None = None, window="hann", center: bool = True,
                   pad_mode: str = "constant", power: float = 2.0, n_mels: int = 128,
                   fmin: float = 0.0, fmax: float | None = None, htk: bool = False,
                   norm: str | None = "slaney", _max_key: torch.Tensor | None = None) -> torch.Tensor:
    """mel_basis @ |stft(y)|**power (reference mel.py:245-352), fused on the GPU.

    Returns (n_mels, n_frames) or (batch, n_mels, n_frames) float32.  ``_max_key`` (internal,
    mfcc): a 1-element int32 device tensor that receives the order-preserving key of max(out)."""
    hop_length, win_length = _resolve_stft_args(n_fft, hop_length, win_length)
    y = _x.to_device_f32(y)
    one_d = y.ndim == 1
    if one_d:
        y = y[None, :]
    if y.ndim != 2:
        raise ValueError(f"y must be 1D or 2D, got {y.ndim}D")
    B, L = y.shape
    dev = y.device
    win = _get_padded_window(window, win_length, n_fft, dev)
    T = _frame_count(L, n_fft, hop_length, center, pad_mode)
    fb, plan, desc = _mel_filterbank_full(sr, n_fft, n_mels, fmin, fmax, htk, norm, dev)
    out = torch.empty((B, n_mels, T), dtype=torch.float32, device=dev)
    if B > 0 and L > 0:
        tw = _get_twiddles(n_fft, dev)
        _x.check(_x.dlib(dev).ap_melspec_max_f32(
            _x.ptr(y), B, L, int(n_fft), hop_length, _x.ptr(win), _x.ptr(tw), int(bool(center)),
            _x.PAD_MODES[pad_mode], T, _x.ptr(fb), _x.ptr(plan), desc.ctypes.data, int(n_mels),
            float(power), _x.ptr(out), None if _max_key is None else _max_key.data_ptr(),
            _x.stream_ptr(dev)))
    else:
        out.zero_()
        _max_key = None
    return out[0] if one_d else out
