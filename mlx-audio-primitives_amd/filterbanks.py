"""Linear- and Bark-scale filterbanks — same API as
/root/reference/mlx_audio_primitives/filterbanks.py.  Host NumPy float64 like
the reference (and like mel_filterbank): the banks are tiny, cached, and feed the same plan-based
contraction kernels as the mel bank through ``filterbank_spectrogram``.
"""

from __future__ import annotations

from functools import lru_cache

import numpy as np
import torch

from ._validation import validate_non_negative, validate_positive
from .mel import _triangles
from .windows import _default_device

_BARK_FORMULAS = ("zwicker", "traunmuller")


def _check_formula(formula: str) -> None:
    if formula not in _BARK_FORMULAS:
        raise ValueError(f"Unknown formula: '{formula}'. Supported: 'zwicker', 'traunmuller'")


def hz_to_bark(frequencies, formula: str = "zwicker") -> np.ndarray:
    """Hz -> Bark (reference filterbanks.py:17-55).  zwicker: 13 atan(0.00076 f) + 3.5 atan((f/7500)^2)
    (Zwicker & Terhardt 1980); traunmuller: 26.81 f / (1960 + f) - 0.53 with the two end corrections
    (below 2 Bark: + 0.15 (2 - z); above 20.1 Bark: + 0.22 (z - 20.1))."""
    _check_formula(formula)
    f = np.asarray(frequencies)
    if formula == "zwicker":
        return 13.0 * np.arctan(0.00076 * f) + 3.5 * np.arctan((f / 7500.0) ** 2)
    z = (26.81 * f) / (1960.0 + f) - 0.53
    z = np.where(z < 2, z + 0.15 * (2 - z), z)
    return np.where(z > 20.1, z + 0.22 * (z - 20.1), z)


def bark_to_hz(bark, formula: str = "zwicker") -> np.ndarray:
    """Bark -> Hz (reference filterbanks.py:58-106).  The Zwicker formula has no closed inverse: five
    Newton steps from 600 sinh(z / 6), forward-difference slope with step 1e-6 Hz floored at 1e-10,
    iterate clipped at 0 - the reference's iteration, so the corner frequencies agree to the bit."""
    _check_formula(formula)
    z = np.asarray(bark)
    if formula == "traunmuller":
        z = np.asarray(z, dtype=np.float64)
        z = np.where(z < 2, z - 0.15 * (2 - z) / 1.15, z)
        z = np.where(z > 20.1, z - 0.22 * (z - 20.1) / 1.22, z)
        return 1960.0 * (z + 0.53) / (26.28 - z)
    hz = 600.0 * np.sinh(z / 6.0)
    step = 1e-6
    for _ in range(5):
        here = hz_to_bark(hz, formula="zwicker")
        slope = np.maximum((hz_to_bark(hz + step, formula="zwicker") - here) / step, 1e-10)
        hz = np.maximum(hz - (here - z) / slope, 0)
    return hz


def _check_band_args(n_bands, fmin, fmax, sr):
    validate_positive(n_bands, "n_bands")
    validate_non_negative(fmin, "fmin")
    if fmax is None:
        fmax = sr / 2.0
    if fmin >= fmax:
        raise ValueError(f"fmin ({fmin}) must be less than fmax ({fmax})")
    if fmax > sr / 2.0:
        raise ValueError(f"fmax ({fmax}) cannot exceed Nyquist frequency ({sr / 2.0})")
    return fmax


@lru_cache(maxsize=64)
def _bark_bank(sr, n_fft, n_bands, fmin, fmax, formula, norm) -> np.ndarray:
    lo = hz_to_bark(np.array([fmin]), formula=formula)[0]
    hi = hz_to_bark(np.array([fmax]), formula=formula)[0]
    return _triangles(bark_to_hz(np.linspace(lo, hi, n_bands + 2), formula=formula), sr, n_fft, norm)


@lru_cache(maxsize=64)
def _linear_bank(sr, n_fft, n_bands, fmin, fmax, norm) -> np.ndarray:
    return _triangles(np.linspace(fmin, fmax, n_bands + 2), sr, n_fft, norm)


_device_cache: dict[tuple, torch.Tensor] = {}


def _on_device(key, bank: np.ndarray, device) -> torch.Tensor:
    dev = _default_device(device)
    key = key + (str(dev),)
    t = _device_cache.get(key)
    if t is None:
        t = torch.from_numpy(bank.copy()).to(dev)
        _device_cache[key] = t
    return t.clone()          # callers own what they get: an in-place edit must not reach the cache


def bark_filterbank(sr: int, n_fft: int, n_bands: int = 24, fmin: float = 0.0, fmax: float | None = None,
                    formula: str = "zwicker", norm: str | None = "slaney", device=None) -> torch.Tensor:
    """Bark-scale filterbank (n_bands, n_fft//2+1), float32, cached (reference filterbanks.py:163-232)."""
    fmax = _check_band_args(n_bands, fmin, fmax, sr)
    _check_formula(formula)
    return _on_device(("bark", sr, n_fft, n_bands, fmin, fmax, formula, norm),
                      _bark_bank(sr, n_fft, n_bands, fmin, fmax, formula, norm), device)


def linear_filterbank(sr: int, n_fft: int, n_bands: int = 64, fmin: float = 0.0, fmax: float | None = None,
                      norm: str | None = "slaney", device=None) -> torch.Tensor:
    """Equal-width (Hz) filterbank (n_bands, n_fft//2+1), float32, cached (reference
    filterbanks.py:273-342)."""
    fmax = _check_band_args(n_bands, fmin, fmax, sr)
    return _on_device(("linear", sr, n_fft, n_bands, fmin, fmax, norm),
                      _linear_bank(sr, n_fft, n_bands, fmin, fmax, norm), device)
