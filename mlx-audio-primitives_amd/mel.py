"""Mel scale, mel filterbank and mel spectrogram — same API as
/root/reference/mlx_audio_primitives/mel.py.

``hz_to_mel`` / ``mel_to_hz`` / the filterbank are host NumPy float64 exactly as in
the reference (mel.py:31-168; the reference keeps these off the device on
purpose).  ``melspectrogram`` is ONE fused kernel launch: the reference's
stft -> abs -> power -> matmul chain (mel.py:310-350) with no (B,F,T) intermediate.
"""

from __future__ import annotations

from functools import lru_cache

import hashlib

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


def _triangles(corners: np.ndarray, sr, n_fft: int, norm) -> np.ndarray:
    """Dense (len(corners) - 2, n_fft//2+1) float32 bank of triangles over the FFT bin centres with
    corner frequencies `corners` (Hz, ascending): filter i rises from corners[i] to its peak at
    corners[i+1] and falls to corners[i+2].  Bit-compatible with the reference's NumPy builders
    (mel.py:100-168, filterbanks.py:113-160,226-265), whose two quirks are part of the contract: each
    ramp's denominator carries a +1e-10 guard, and the triangle is rounded to float32 BEFORE the
    Slaney area normalisation 2 / (f[i+2] - f[i]) (a float64 product rounded to float32 once more)."""
    if norm not in ("slaney", None):
        raise ValueError(f"Unknown norm: '{norm}'. Supported: 'slaney', None")
    left, peak, right = corners[:-2], corners[1:-1], corners[2:]
    grid = np.linspace(0, sr / 2.0, 1 + n_fft // 2)[np.newaxis, :]
    rising = (grid - left[:, np.newaxis]) / (peak - left + 1e-10)[:, np.newaxis]
    falling = (right[:, np.newaxis] - grid) / (right - peak + 1e-10)[:, np.newaxis]
    bank = np.maximum(0, np.minimum(rising, falling)).astype(np.float32)
    if norm == "slaney":
        np.multiply(bank, (2.0 / (right - left))[:, np.newaxis], out=bank, casting="same_kind")
    bank.setflags(write=False)
    return bank


@lru_cache(maxsize=64)
def _triangular_bank(sr, n_fft, n_mels, fmin, fmax, htk, norm) -> np.ndarray:
    """Mel filterbank: triangles whose corners are equally spaced on the mel scale (mel.py:100-168)."""
    corners = mel_to_hz(np.linspace(hz_to_mel(fmin, htk=htk), hz_to_mel(fmax, htk=htk), n_mels + 2),
                        htk=htk)
    return _triangles(corners, sr, n_fft, norm)


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
                   norm: str | None = "slaney") -> torch.Tensor:
    """mel_basis @ |stft(y)|**power (reference mel.py:245-352), fused on the GPU.

    Returns (n_mels, n_frames) or (batch, n_mels, n_frames) float32."""
    return _melspectrogram_max(y, sr, n_fft, hop_length, win_length, window, center, pad_mode, power, n_mels,
                               fmin, fmax, htk, norm, None, lines=True)


def _melspectrogram_max(y, sr, n_fft, hop_length, win_length, window, center, pad_mode, power, n_mels, fmin, fmax,
                        htk, norm, max_key, lines=False):
    """`melspectrogram` that also leaves the order-preserving key of max(out) in `max_key` (a 1-element int32
    device tensor, or None): the reference level of mfcc's dB stage, out of the same kernel.  `lines`: the
    result may be a view with padded rows (see stft._SPECTRUM_LAYOUT); internal callers that hand the array
    to a kernel that expects dense rows leave it off."""
    def bank(dev):
        return _mel_filterbank_full(sr, n_fft, n_mels, fmin, fmax, htk, norm, dev)

    return _bank_spectrogram(y, n_fft, hop_length, win_length, window, center, pad_mode, power, bank, max_key, lines)


def _bank_spectrogram(y, n_fft, hop_length, win_length, window, center, pad_mode, power, bank, max_key=None,
                      lines=False):
    """bank(dev) -> (dense filterbank (M, F), contraction plan, plan descriptor) on `dev`;
    returns bank @ |stft(y)|**power from ONE fused kernel launch."""
    hop_length, win_length = _resolve_stft_args(n_fft, hop_length, win_length)
    pcm16 = _is_pcm16(y)
    y = _to_device_pcm16(y) if pcm16 else _x.to_device_f32(y)
    one_d = y.ndim == 1
    if one_d:
        y = y[None, :]
    if y.ndim != 2:
        raise ValueError(f"y must be 1D or 2D, got {y.ndim}D")
    B, L = y.shape
    dev = y.device
    win = _get_padded_window(window, win_length, n_fft, dev)
    T = _frame_count(L, n_fft, hop_length, center, pad_mode)
    fb, plan, desc = bank(dev)
    n_rows = fb.shape[0]
    # n_fft = 2048 run kernel: rows padded to a multiple of 8 frames make every 8-frame output run a whole aligned
    # 32-byte sector (stft._SPECTRUM_LAYOUT: the result is then a strided view with the dense layout's values)
    from .stft import _spectrum_layout
    Ts = T
    if lines and not pcm16 and _spectrum_layout() == "lines" and T % 8 and B * T >= 512 and \
            _x.lib().ap_melspec_rows_fused(int(n_fft), hop_length, int(bool(center)), _x.PAD_MODES[pad_mode],
                                          int(n_rows), float(power), _x.ptr(plan), desc.ctypes.data):
        Ts = -(-T // 8) * 8
    out = torch.empty((B, n_rows, Ts), dtype=torch.float32, device=dev)
    if B > 0 and L > 0:
        tw = _get_twiddles(n_fft, dev)
        key_ptr = None if max_key is None else max_key.data_ptr()
        if Ts != T:
            rc = _x.dlib(dev).ap_melspec_rows_f32(
                _x.ptr(y), B, L, int(n_fft), hop_length, _x.ptr(win), _x.ptr(tw), int(bool(center)),
                _x.PAD_MODES[pad_mode], T, Ts, _x.ptr(fb), _x.ptr(plan), desc.ctypes.data, int(n_rows),
                float(power), _x.ptr(out), key_ptr, _x.stream_ptr(dev))
            if rc == _x.AP_ERR_UNSUPPORTED:          # a shape the run kernel turns down after all: dense rows
                Ts = T
                out = torch.empty((B, n_rows, T), dtype=torch.float32, device=dev)
            else:
                _x.check(rc)
        if Ts != T:
            pass
        elif pcm16:
            # 16-bit PCM: converted inside the n_fft=2048 run kernel's loads where
            # that kernel applies, else by one conversion pass into a float32 scratch copy
            fused = _x.lib().ap_melspec_pcm16_fused(L, int(n_fft), hop_length, int(bool(center)),
                                                    _x.PAD_MODES[pad_mode], int(n_rows), float(power),
                                                    desc.ctypes.data)
            fused = fused and y.data_ptr() % 4 == 0      # the fused loads read sample pairs as dwords
            scratch = None if fused else torch.empty((B, L), dtype=torch.float32, device=dev)
            _x.check(_x.dlib(dev).ap_melspec_pcm16_f32(
                _x.ptr(y), B, L, int(n_fft), hop_length, _x.ptr(win), _x.ptr(tw), int(bool(center)),
                _x.PAD_MODES[pad_mode], T, _x.ptr(fb), _x.ptr(plan), desc.ctypes.data, int(n_rows),
                float(power), _x.ptr(out), key_ptr, None if scratch is None else _x.ptr(scratch),
                _x.stream_ptr(dev)))
        else:
            _x.check(_x.dlib(dev).ap_melspec_max_f32(
                _x.ptr(y), B, L, int(n_fft), hop_length, _x.ptr(win), _x.ptr(tw), int(bool(center)),
                _x.PAD_MODES[pad_mode], T, _x.ptr(fb), _x.ptr(plan), desc.ctypes.data, int(n_rows),
                float(power), _x.ptr(out), key_ptr, _x.stream_ptr(dev)))
    else:
        out.zero_()
    if Ts != T:
        out = out[:, :, :T]
    return out[0] if one_d else out


def _is_pcm16(y) -> bool:
    return (isinstance(y, torch.Tensor) and y.dtype == torch.int16) or \
        (isinstance(y, np.ndarray) and y.dtype == np.int16)


def _to_device_pcm16(y) -> torch.Tensor:
    """Contiguous int16 tensor in HBM (full scale +-32768 = +-1.0)."""
    if isinstance(y, torch.Tensor):
        dev = y.device if y.is_cuda else _x.require_device()
        _x.lib()
        return y.to(dev).contiguous()
    return torch.from_numpy(np.ascontiguousarray(y)).to(_x.require_device())


def pcm16_to_float(y) -> torch.Tensor:
    """int16 PCM -> float32 in [-1, 1) on the device (x / 32768): the ingest step for every operator
    other than the mel front end, which converts inside its own loads."""
    y16 = _to_device_pcm16(y)
    out = torch.empty(y16.shape, dtype=torch.float32, device=y16.device)
    if y16.numel():
        _x.check(_x.dlib(y16.device).ap_pcm16_to_f32(_x.ptr(y16), y16.numel(), 1.0 / 32768.0, _x.ptr(out),
                                                     _x.stream_ptr(y16.device)))
    return out


_custom_bank_cache: dict[tuple, tuple] = {}


def filterbank_spectrogram(y, filterbank, n_fft: int = 2048, hop_length: int | None = None,
                           win_length: int | None = None, window="hann", center: bool = True,
                           pad_mode: str = "constant", power: float = 2.0) -> torch.Tensor:
    """filterbank @ |stft(y)|**power for ANY (n_bands, n_fft//2+1) bank - e.g. ``bark_filterbank`` or
    ``linear_filterbank`` - through the same fused kernels as ``melspectrogram`` (not in the reference,
    which stops at building those banks, filterbanks.py:163-342)."""
    fbt = filterbank if isinstance(filterbank, torch.Tensor) else torch.as_tensor(np.asarray(filterbank))
    fb_np = np.ascontiguousarray(fbt.detach().cpu().numpy(), dtype=np.float32)
    if fb_np.ndim != 2 or fb_np.shape[1] != 1 + n_fft // 2:
        raise ValueError(f"filterbank must have shape (n_bands, {1 + n_fft // 2}), got {tuple(fb_np.shape)}")

    def bank(dev):
        key = (hashlib.sha256(fb_np.tobytes()).digest(), fb_np.shape, str(dev))   # content-addressed: no collisions
        hit = _custom_bank_cache.get(key)
        if hit is None:
            plan_np, desc = _x.mel_plan_host(fb_np)
            hit = (torch.from_numpy(fb_np.copy()).to(dev), torch.from_numpy(plan_np).to(dev), desc)
            if len(_custom_bank_cache) >= 32:
                _custom_bank_cache.pop(next(iter(_custom_bank_cache)))
            _custom_bank_cache[key] = hit
        return hit

    return _bank_spectrogram(y, n_fft, hop_length, win_length, window, center, pad_mode, power, bank)
