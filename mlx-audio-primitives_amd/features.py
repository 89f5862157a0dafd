"""Spectral features — same API as /root/reference/mlx_audio_primitives/features.py
(spectral_centroid / spectral_bandwidth / spectral_rolloff / spectral_flatness /
zero_crossing_rate).

From audio at n_fft = 2048 (the default) ONE kernel goes from the samples to the statistics - the
spectrum is never written (csrc/kernels_mel2048.h, ap_spec2048_run_kernel).  Other shapes chain the fused
STFT kernel and ONE statistics kernel that takes |X| (and |X|**power) on load from the complex
spectrum — the reference's magnitude / power / sum / cumsum / argmax tensors
(features.py:24-55,115-134,342-360) are never materialised.  From a given
spectrogram S the same kernel reads S once.  ``spectral_contrast`` (a host NumPy sort per octave band
in the reference, features.py:445-595) is one selection kernel over the octave bands.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _extension as _x
from ._validation import validate_positive, validate_range
from .stft import stft

_freq_cache: dict[tuple, torch.Tensor] = {}


def _get_frequencies(sr: int, n_fft: int, device) -> torch.Tensor:
    """Bin centres linspace(0, sr/2, n_fft//2+1), float32 (reference features.py:19-21)."""
    key = (sr, n_fft, str(device))
    t = _freq_cache.get(key)
    if t is None:
        t = torch.from_numpy(np.linspace(0, sr / 2.0, n_fft // 2 + 1).astype(np.float32)).to(device)
        _freq_cache[key] = t
    return t


def _spectral_from_audio(y, sr, n_fft, hop_length, win_length, window, center, pad_mode, freq, power, want,
                         centroid, p, norm, roll_percent, amin):
    """One kernel from the samples to the statistics (n_fft = 2048, constant padding): the complex
    spectrum is never written.  Returns None when the shape is not served (the caller then runs the
    fused STFT kernel + the statistics kernel)."""
    from .stft import _frame_count, _get_padded_window, _get_twiddles, _resolve_stft_args

    if centroid is not None or n_fft != 2048:
        return None
    hop_length, win_length = _resolve_stft_args(n_fft, hop_length, win_length)
    y = _x.to_device_f32(y)
    one_d = y.ndim == 1
    if one_d:
        y = y[None, :]
    if y.ndim != 2 or y.shape[0] == 0 or y.shape[1] == 0:
        return None
    B, L = y.shape
    if pad_mode not in _x.PAD_MODES:
        return None
    if not _x.lib().ap_spectral_audio_fused(L, int(n_fft), int(hop_length), int(bool(center)), _x.PAD_MODES[pad_mode]):
        return None
    dev = y.device
    F = n_fft // 2 + 1
    freq = _get_frequencies(sr, n_fft, dev) if freq is None else _x.to_device_f32(freq, dev)
    if freq.ndim != 1 or freq.shape[0] != F:
        raise ValueError(f"freq must be 1D with {F} entries (freq_bins), got shape {tuple(freq.shape)}")
    T = _frame_count(L, n_fft, hop_length, center, pad_mode)
    win = _get_padded_window(window, win_length, n_fft, dev)
    tw = _get_twiddles(n_fft, dev)
    outs = {k: torch.empty((B, 1, T), dtype=torch.float32, device=dev) for k in want}
    g = lambda k: _x.ptr(outs[k]) if k in outs else None  # noqa: E731
    _x.check(_x.dlib(dev).ap_spectral_audio_f32(
        _x.ptr(y.contiguous()), B, L, int(n_fft), int(hop_length), _x.ptr(win), _x.ptr(tw), int(bool(center)),
        _x.PAD_MODES[pad_mode], T, _x.ptr(freq.contiguous()), float(power), float(p), int(bool(norm)),
        float(roll_percent), float(amin), g("centroid"), g("bandwidth"), g("rolloff"), g("flatness"),
        _x.stream_ptr(dev)))
    return {k: (v[0] if one_d else v) for k, v in outs.items()}


def _spectral(y, S, sr, n_fft, hop_length, win_length, window, center, pad_mode, freq, power=1.0,
              want=(), centroid=None, p=2.0, norm=True, roll_percent=0.85, amin=1e-10):
    """Run the statistics kernel; returns {name: (1,T) | (B,1,T) tensor} for the names in `want`."""
    if S is not None:
        S = _x.to_device_f32(S)
        is_complex = 0
        power = 1.0                       # a given S is used as it is (features.py:36-37)
    else:
        if y is None:
            raise ValueError("Either y (audio) or S (spectrogram) must be provided")
        fused = _spectral_from_audio(y, sr, n_fft, hop_length, win_length, window, center, pad_mode, freq, power,
                                     want, centroid, p, norm, roll_percent, amin)
        if fused is not None:
            return fused
        Sc = stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                  center=center, pad_mode=pad_mode)
        S = torch.view_as_real(Sc)        # (…, F, T, 2): |X| is taken inside the kernel
        is_complex = 1
    nd = S.ndim - is_complex
    if nd not in (2, 3):
        raise ValueError("S must be 2D (freq_bins, n_frames) or 3D (batch, freq_bins, n_frames)")
    batched = nd == 3
    if not batched:
        S = S[None]
    S = S.contiguous()
    B, F, T = S.shape[0], S.shape[1], S.shape[2]
    dev = S.device
    if freq is None:
        freq = _get_frequencies(sr, n_fft, dev)
    else:
        freq = _x.to_device_f32(freq, dev)
    if freq.ndim != 1 or freq.shape[0] != F:
        raise ValueError(f"freq must be 1D with {F} entries (freq_bins), got shape {tuple(freq.shape)}")
    cin = None
    if centroid is not None:
        cin = _x.to_device_f32(centroid, dev).reshape(-1)
        if cin.numel() != B * T:
            cin = cin.reshape(-1).expand(B * T) if cin.numel() == 1 else None
            if cin is None:
                raise ValueError("centroid must have shape (1, n_frames) or (batch, 1, n_frames)")
        cin = cin.contiguous()
    outs = {k: torch.empty((B, 1, T), dtype=torch.float32, device=dev) for k in want}
    if B > 0 and T > 0:
        g = lambda k: _x.ptr(outs[k]) if k in outs else None  # noqa: E731
        _x.check(_x.dlib(dev).ap_spectral_stats_f32(
            _x.ptr(S), is_complex, B, F, T, _x.ptr(freq), float(power), None if cin is None else _x.ptr(cin),
            float(p), int(bool(norm)), float(roll_percent), float(amin), g("centroid"), g("bandwidth"),
            g("rolloff"), g("flatness"), _x.stream_ptr(dev)))
    return {k: (v if batched else v[0]) for k, v in outs.items()}


def spectral_centroid(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                      win_length: int | None = None, window="hann", center: bool = True,
                      pad_mode: str = "constant", freq=None) -> torch.Tensor:
    """sum(f S) / sum(S) per frame (reference features.py:57-134).  (1, T) or (batch, 1, T)."""
    return _spectral(y, S, sr, n_fft, hop_length, win_length, window, center, pad_mode, freq,
                     want=("centroid",))["centroid"]


def spectral_bandwidth(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                       win_length: int | None = None, window="hann", center: bool = True,
                       pad_mode: str = "constant", freq=None, centroid=None, p: float = 2.0,
                       norm: bool = True) -> torch.Tensor:
    """(sum(S |f - centroid|^p) / sum(S))^(1/p) per frame (reference features.py:137-271)."""
    return _spectral(y, S, sr, n_fft, hop_length, win_length, window, center, pad_mode, freq,
                     want=("bandwidth",), centroid=centroid, p=p, norm=norm)["bandwidth"]


def spectral_rolloff(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                     win_length: int | None = None, window="hann", center: bool = True,
                     pad_mode: str = "constant", freq=None, roll_percent: float = 0.85,
                     use_cpp: bool = True) -> torch.Tensor:
    """Frequency below which roll_percent of the frame's energy lies (reference
    features.py:274-360, native spectral.cpp:125-207).  ``use_cpp`` is accepted for signature
    compatibility; there is one (device) implementation."""
    validate_range(roll_percent, "roll_percent", min_val=0.0, max_val=1.0)
    return _spectral(y, S, sr, n_fft, hop_length, win_length, window, center, pad_mode, freq,
                     want=("rolloff",), roll_percent=roll_percent)["rolloff"]


def spectral_flatness(y=None, S=None, n_fft: int = 2048, hop_length: int = 512,
                      win_length: int | None = None, window="hann", center: bool = True,
                      pad_mode: str = "constant", power: float = 2.0, amin: float = 1e-10) -> torch.Tensor:
    """Geometric mean / arithmetic mean of max(|X|**power, amin) (reference features.py:363-442)."""
    return _spectral(y, S, 22050, n_fft, hop_length, win_length, window, center, pad_mode,
                     _flat_freq(S, n_fft), power=power, want=("flatness",), amin=amin)["flatness"]


def _flat_freq(S, n_fft):
    """spectral_flatness takes no sr / freq: any bin-centre table of the right length will do."""
    if S is None:
        return None
    F = S.shape[-2]
    return np.zeros(F, np.float32)


def spectral_features(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                      win_length: int | None = None, window="hann", center: bool = True,
                      pad_mode: str = "constant", freq=None, p: float = 2.0, norm: bool = True,
                      roll_percent: float = 0.85) -> dict:
    """centroid, bandwidth and rolloff of every frame from ONE pass over the spectrum (not in the
    reference, where three calls recompute the STFT three times, features.py:112-114)."""
    validate_range(roll_percent, "roll_percent", min_val=0.0, max_val=1.0)
    return _spectral(y, S, sr, n_fft, hop_length, win_length, window, center, pad_mode, freq,
                     want=("centroid", "bandwidth", "rolloff"), p=p, norm=norm, roll_percent=roll_percent)


def zero_crossing_rate(y, frame_length: int = 2048, hop_length: int = 512, center: bool = True,
                       pad_mode: str = "edge", use_mlx: bool = True) -> torch.Tensor:
    """Fraction of sign changes per frame (reference features.py:625-722; ``use_mlx`` is accepted
    for signature compatibility — the device kernel follows the default path, (x >= 0) sign tests)."""
    from .framing import _frame_stats

    return _frame_stats(y, frame_length, hop_length, center, pad_mode, "zcr")


def _contrast_bands(freq: np.ndarray, fmin: float, n_bands: int, quantile: float) -> np.ndarray:
    """(n_bands + 1, 3) int32: first bin, one past the last bin and the number of extreme values of every
    octave band [0, fmin], [fmin, 2 fmin], ... (librosa's rule as the reference applies it, features.py:528-561:
    the neighbour bin below joins every band but the first, the last band runs to Nyquist, the count comes from
    the band BEFORE its top bin is dropped, every band but the last drops its top bin)."""
    edges = np.zeros(n_bands + 2)
    edges[1:] = fmin * (2.0 ** np.arange(0, n_bands + 1))
    F = freq.shape[0]
    rows = np.zeros((n_bands + 1, 3), np.int32)
    for k in range(n_bands + 1):
        inside = np.flatnonzero((freq >= edges[k]) & (freq <= edges[k + 1]))
        if inside.size == 0:
            continue
        lo, hi = int(inside[0]), int(inside[-1]) + 1
        if k > 0 and lo > 0:
            lo -= 1
        if k == n_bands:
            hi = F
        count = hi - lo
        n_q = int(max(np.rint(quantile * count), 1))
        if k < n_bands and count > 1:
            hi -= 1
        rows[k] = (lo, hi, n_q)
    return rows


def spectral_contrast(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                      win_length: int | None = None, window="hann", center: bool = True,
                      pad_mode: str = "constant", freq=None, fmin: float = 200.0, n_bands: int = 6,
                      quantile: float = 0.02, linear: bool = False) -> torch.Tensor:
    """Peak-to-valley contrast of every octave band and frame (reference features.py:445-595):
    (n_bands + 1, n_frames) or (batch, n_bands + 1, n_frames)."""
    validate_positive(n_bands, "n_bands")
    validate_range(quantile, "quantile", min_val=0.0, max_val=1.0)
    if S is not None:
        S = _x.to_device_f32(S)
    else:
        if y is None:
            raise ValueError("Either y (audio) or S (spectrogram) must be provided")
        from .stft import magnitude
        S = magnitude(stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                           center=center, pad_mode=pad_mode))
    if S.ndim not in (2, 3):
        raise ValueError("S must be 2D (freq_bins, n_frames) or 3D (batch, freq_bins, n_frames)")
    batched = S.ndim == 3
    if not batched:
        S = S[None]
    S = S.contiguous()
    B, F, T = S.shape
    dev = S.device
    if freq is None:
        fr = np.linspace(0, sr / 2.0, n_fft // 2 + 1).astype(np.float32)
    else:
        fr = np.asarray(freq.detach().cpu() if isinstance(freq, torch.Tensor) else freq, dtype=np.float32).reshape(-1)
    if fr.shape[0] != F:
        raise ValueError(f"freq must be 1D with {F} entries (freq_bins), got shape {tuple(fr.shape)}")
    bands = _contrast_bands(fr, float(fmin), int(n_bands), float(quantile))
    out = torch.empty((B, n_bands + 1, T), dtype=torch.float32, device=dev)
    if B > 0 and T > 0:
        bd = torch.from_numpy(bands).to(dev)
        _x.check(_x.dlib(dev).ap_spectral_contrast_f32(_x.ptr(S), B, F, T, _x.ptr(bd), n_bands + 1, int(bool(linear)),
                                                       _x.ptr(out), _x.stream_ptr(dev)))
    return out if batched else out[0]
