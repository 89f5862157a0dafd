"""STFT / ISTFT — same API as /root/reference/mlx_audio_primitives/stft.py.

The reference builds pad -> frame -> window -> rfft -> transpose out of five ops
(stft.py:109-133,216) and irfft -> overlap-add out of three (stft.py:292-312).
Here each direction is one call into libaudioprims_hip.so; this file only keeps
the reference's argument handling, error messages, caches and length logic.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _extension as _x
from .windows import get_window

_WINDOW_SUM_EPSILON = 1e-8      # stft.py:21 (applied inside the overlap-add kernel)
_WINDOW_CACHE_MAXSIZE = 32      # stft.py:24


class _WindowCache:
    """LRU of centre-padded windows per device, content-hashed for array windows
    (reference stft.py:27-81)."""

    def __init__(self, maxsize: int = _WINDOW_CACHE_MAXSIZE):
        self._cache: dict[tuple, torch.Tensor] = {}
        self._order: list[tuple] = []
        self._maxsize = maxsize

    @staticmethod
    def _key(window, win_length, n_fft, device):
        if isinstance(window, str):
            return (window, win_length, n_fft, str(device))
        w = window.detach().cpu().numpy() if isinstance(window, torch.Tensor) else np.asarray(window)
        return ("array", hash(np.ascontiguousarray(w, dtype=np.float32).tobytes()), win_length,
                n_fft, str(device))

    def get(self, key):
        if key in self._cache:
            self._order.remove(key)
            self._order.append(key)
            return self._cache[key]
        return None

    def put(self, key, value):
        while len(self._cache) >= self._maxsize and self._order:
            self._cache.pop(self._order.pop(0), None)
        self._cache[key] = value
        self._order.append(key)

    def clear(self):
        self._cache.clear()
        self._order.clear()


_padded_window_cache = _WindowCache()
_twiddle_cache: dict[tuple, torch.Tensor] = {}


def _get_padded_window(window, win_length: int, n_fft: int, device) -> torch.Tensor:
    """Window centre-padded to n_fft (reference stft.py:88-106)."""
    key = _WindowCache._key(window, win_length, n_fft, device)
    hit = _padded_window_cache.get(key)
    if hit is not None:
        return hit
    win = get_window(window, win_length, fftbins=True, device=device).to(device)
    if win_length < n_fft:
        left = (n_fft - win_length) // 2
        right = n_fft - win_length - left
        win = torch.nn.functional.pad(win, (left, right))
    win = win.contiguous()
    _padded_window_cache.put(key, win)
    return win


def _get_twiddles(n_fft: int, device) -> torch.Tensor:
    key = (n_fft, str(device))
    t = _twiddle_cache.get(key)
    if t is None:
        t = torch.from_numpy(_x.twiddle_table_host(n_fft)).to(device)
        _twiddle_cache[key] = t
    return t


def _resolve_stft_args(n_fft, hop_length, win_length):
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if win_length <= 0:
        raise ValueError(f"win_length must be positive, got {win_length}")
    if win_length > n_fft:
        raise ValueError(f"win_length ({win_length}) must be <= n_fft ({n_fft})")
    if hop_length > n_fft:
        raise ValueError(
            f"hop_length ({hop_length}) should typically be <= n_fft ({n_fft})"
        )
    return int(hop_length), int(win_length)


def _frame_count(signal_length: int, n_fft: int, hop_length: int, center: bool, pad_mode: str):
    if pad_mode not in _x.PAD_MODES:
        raise ValueError(
            f"Unknown pad_mode: '{pad_mode}'. Supported: reflect, constant, edge"
        )
    padded = signal_length + (2 * (n_fft // 2) if center else 0)
    if padded < n_fft:
        raise ValueError(
            f"Signal length ({padded}) must be >= frame_length ({n_fft}). "
            f"Consider padding the signal."
        )
    return 1 + (padded - n_fft) // hop_length


# Layout of the spectrum `stft` returns for n_fft = 2048 / 512 / 400 / 256.  "lines" (default): the (…, F, T) result is a strided
# VIEW of a buffer whose rows are padded to a multiple of 16 frames, so every row starts on a 128-byte line and
# the kernel writes whole lines (0.28 instead of 0.36 ms on 256 x 10 s; `istft`, `magnitude`, `phase` and
# `griffinlim` read such views in place).  The values, shape and dtype are the reference's; only
# `.is_contiguous()` differs - as it does for the reference itself, whose result is the transposed view
# mx.transpose(…, (0, 2, 1)) of its (B, T, F) transform (stft.py:216).  "dense": a contiguous array
# (what the C entry point ap_stft_f32 always writes).  AP_SPECTRUM_LAYOUT=dense in the environment sets the default.
import os as _os

_SPECTRUM_LAYOUT = "dense" if _os.environ.get("AP_SPECTRUM_LAYOUT", "lines") == "dense" else "lines"
_LINES_N_FFT = (2048, 512, 400, 256)       # the kernels that write / read padded rows (ap_stft_rows_f32, ap_istft_rows_f32)


def _spectrum_layout() -> str:
    return _SPECTRUM_LAYOUT


def set_spectrum_layout(layout: str) -> str:
    """Choose "lines" (rows padded to whole 128-byte lines, default) or "dense" for `stft`'s n_fft = 2048
    results; returns the previous setting."""
    global _SPECTRUM_LAYOUT
    if layout not in ("lines", "dense"):
        raise ValueError(f"Unknown spectrum layout: '{layout}'. Supported: 'lines', 'dense'")
    prev, _SPECTRUM_LAYOUT = _SPECTRUM_LAYOUT, layout
    return prev


def stft(y, n_fft: int = 2048, hop_length: int | None = None, win_length: int | None = None,
         window="hann", center: bool = True, pad_mode: str = "constant") -> torch.Tensor:
    """Short-time Fourier transform (reference stft.py:136-222).

    y: (samples,) or (batch, samples).  Returns complex64 (n_fft//2+1, n_frames) or
    (batch, n_fft//2+1, n_frames), librosa layout."""
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
    F = n_fft // 2 + 1
    Ts = T
    if n_fft in _LINES_N_FFT and _SPECTRUM_LAYOUT == "lines" and T % 16 and B > 0 and L > 0 and B * T >= 512 \
            and T < 490000 - 16:          # (the n_fft = 2048 ISTFT addresses a clip as one buffer resource: 1025 Ts 8 < 0xF0000000 bytes)
        Ts = -(-T // 16) * 16                # rows padded to whole 128-byte lines (see _SPECTRUM_LAYOUT)
    out = torch.empty((B, F, Ts, 2), dtype=torch.float32, device=dev)
    if B > 0 and L > 0:
        tw = _get_twiddles(n_fft, dev)
        if Ts != T:
            _x.check(_x.dlib(dev).ap_stft_rows_f32(_x.ptr(y), B, L, int(n_fft), hop_length, _x.ptr(win),
                                                   _x.ptr(tw), int(bool(center)), _x.PAD_MODES[pad_mode], T, Ts,
                                                   _x.ptr(out), _x.stream_ptr(dev)))
        else:
            _x.check(_x.dlib(dev).ap_stft_f32(_x.ptr(y), B, L, int(n_fft), hop_length, _x.ptr(win),
                                          _x.ptr(tw), int(bool(center)), _x.PAD_MODES[pad_mode], T,
                                          _x.ptr(out), _x.stream_ptr(dev)))
    else:
        out.zero_()
    S = torch.view_as_complex(out)
    if Ts != T:
        S = S[:, :, :T]
    return S[0] if one_d else S


def _padded_row_stride(S: torch.Tensor):
    """Row stride (complex values) of a (B, F, T) view whose rows are padded but otherwise dense, else None."""
    if S.ndim != 3 or S.is_contiguous():
        return None
    B, F, T = S.shape
    sb, sf, st = S.stride()
    if T > 0 and st == 1 and sf > T and (B == 1 or sb == F * sf):
        return int(sf)
    return None


def stft_padded_rows(y, n_fft: int = 2048, hop_length: int | None = None, win_length: int | None = None,
                     window="hann", center: bool = True, pad_mode: str = "constant",
                     row_multiple: int = 16, out: torch.Tensor | None = None) -> torch.Tensor:
    """`stft` into a buffer whose rows are padded to a multiple of `row_multiple` frames (n_fft = 2048).

    Returns the (batch, n_fft//2+1, n_frames) result as a strided VIEW of the padded buffer: same
    values as `stft`, but every row starts on a 128-byte line, so the kernel writes whole lines and
    needs no carries (ap_stft_rows_f32; the layout the Griffin-Lim workspaces use).  Not part of the
    reference's API: its `stft` always returns the dense array (stft.py:216)."""
    hop_length, win_length = _resolve_stft_args(n_fft, hop_length, win_length)
    y = _x.to_device_f32(y)
    if y.ndim != 2:
        raise ValueError(f"y must be 2D, got {y.ndim}D")
    B, L = y.shape
    dev = y.device
    win = _get_padded_window(window, win_length, n_fft, dev)
    T = _frame_count(L, n_fft, hop_length, center, pad_mode)
    F = n_fft // 2 + 1
    Ts = -(-T // row_multiple) * row_multiple
    if out is None:                        # the padding columns are never written (nor read by this package)
        out = torch.empty((B, F, Ts, 2), dtype=torch.float32, device=dev)
    elif out.shape != (B, F, Ts, 2) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
        raise ValueError(f"out must be a contiguous float32 tensor of shape {(B, F, Ts, 2)} on {dev}")
    tw = _get_twiddles(n_fft, dev)
    _x.check(_x.dlib(dev).ap_stft_rows_f32(_x.ptr(y), B, L, int(n_fft), hop_length, _x.ptr(win),
                                           _x.ptr(tw), int(bool(center)), _x.PAD_MODES[pad_mode], T, Ts,
                                           _x.ptr(out), _x.stream_ptr(dev)))
    return torch.view_as_complex(out)[:, :, :T]


def istft(stft_matrix, hop_length: int | None = None, win_length: int | None = None,
          n_fft: int | None = None, window="hann", center: bool = True,
          length: int | None = None) -> torch.Tensor:
    """Inverse STFT (reference stft.py:225-344): irfft, windowed overlap-add,
    division by max(sum w^2, 1e-8), centre trim / length fix — one library call."""
    if not isinstance(stft_matrix, torch.Tensor):
        stft_matrix = torch.as_tensor(np.asarray(stft_matrix))
    if stft_matrix.ndim not in (2, 3):
        raise ValueError(f"stft_matrix must be 2D or 3D, got {stft_matrix.ndim}D")
    two_d = stft_matrix.ndim == 2
    if two_d:
        stft_matrix = stft_matrix[None, :]
    dev = stft_matrix.device if stft_matrix.is_cuda else _x.require_device()
    _x.lib()
    S = stft_matrix.to(device=dev, dtype=torch.complex64)
    B, F, T = S.shape
    # rows padded to whole 128-byte lines (stft_padded_rows / ap_stft_rows_f32) are read in place by the
    # fused n_fft = 2048 kernel; every other strided view is made dense first
    row_stride = _padded_row_stride(S)
    if row_stride is None or 2 * (F - 1) not in _LINES_N_FFT or (n_fft is not None and n_fft != 2 * (F - 1)):
        S = S.contiguous()
        row_stride = None
    if n_fft is None:
        n_fft = 2 * (F - 1)
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    if F != n_fft // 2 + 1:
        raise ValueError(
            f"stft_matrix has {F} frequency bins but n_fft={n_fft} needs {n_fft // 2 + 1}"
        )
    win = _get_padded_window(window, win_length, n_fft, dev)
    # output span (reference stft.py:300-338)
    if length is not None:
        padded_length = length + n_fft if center else length
    else:
        padded_length = n_fft + (T - 1) * hop_length
    if center:
        offset = n_fft // 2
        out_len = length if length is not None else max(padded_length - 2 * offset, 0)
        ola_len = out_len
    else:
        offset = 0
        out_len = length if length is not None else padded_length
        # beyond the natural span the reference zero-pads (stft.py:336-338)
        ola_len = min(out_len, padded_length)
    y = torch.zeros((B, out_len), dtype=torch.float32, device=dev) if ola_len < out_len else \
        torch.empty((B, out_len), dtype=torch.float32, device=dev)
    if B > 0 and T > 0 and ola_len > 0:
        tw = _get_twiddles(n_fft, dev)
        # (B, T, n_fft) frames workspace; 0 floats when the fused irfft + overlap-add kernel applies
        n_ws = int(_x.lib().ap_istft_workspace_floats(B, T, int(n_fft), int(hop_length), offset))
        ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
        Sr = torch.view_as_real(S)
        if ola_len == out_len:
            tgt = y
        else:
            tgt = torch.empty((B, ola_len), dtype=torch.float32, device=dev)
        rc = _x.AP_ERR_UNSUPPORTED if row_stride is not None else None
        if row_stride is not None:
            rc = _x.dlib(dev).ap_istft_rows_f32(_x.ptr(Sr), B, T, row_stride, int(n_fft), int(hop_length),
                                                _x.ptr(win), _x.ptr(tw), offset, ola_len, _x.ptr(tgt),
                                                _x.stream_ptr(dev))
            if rc == _x.AP_ERR_UNSUPPORTED:                  # hop the fused kernel does not serve: dense copy
                S = S.contiguous()
                Sr = torch.view_as_real(S)
        if rc is None or rc == _x.AP_ERR_UNSUPPORTED:
            rc = _x.dlib(dev).ap_istft_f32(_x.ptr(Sr), B, T, int(n_fft), int(hop_length), _x.ptr(win),
                                           _x.ptr(tw), _x.ptr(ws), offset, ola_len, _x.ptr(tgt),
                                           _x.stream_ptr(dev))
        _x.check(rc)
        if tgt is not y:
            y[:, :ola_len] = tgt
    return y[0] if two_d else y


def magnitude(stft_matrix) -> torch.Tensor:
    """|S| (reference stft.py:347-362)."""
    return _complex_unary(stft_matrix, "ap_magnitude_f32")


def phase(stft_matrix) -> torch.Tensor:
    """atan2(imag, real) (reference stft.py:365-379)."""
    return _complex_unary(stft_matrix, "ap_phase_f32")


def _complex_unary(S, fn_name: str) -> torch.Tensor:
    if not isinstance(S, torch.Tensor):
        S = torch.as_tensor(np.asarray(S))
    dev = S.device if S.is_cuda else _x.require_device()
    _x.lib()
    S = S.to(device=dev, dtype=torch.complex64)
    out = torch.empty(S.shape, dtype=torch.float32, device=dev)
    row_stride = _padded_row_stride(S)
    if row_stride is not None and S.numel():          # line-padded rows are read in place, the result is dense
        B, F, T = S.shape
        _x.check(_x.dlib(dev).ap_complex_unary_rows_f32(_x.ptr(torch.view_as_real(S)), B * F, T, row_stride,
                                                        0 if fn_name == "ap_magnitude_f32" else 1, _x.ptr(out),
                                                        _x.stream_ptr(dev)))
        return out
    S = S.contiguous()
    n = S.numel()
    if n:
        _x.check(getattr(_x.dlib(dev), fn_name)(_x.ptr(torch.view_as_real(S)), n, _x.ptr(out),
                                            _x.stream_ptr(dev)))
    return out


def check_nola(window, hop_length: int, n_fft: int, tol: float = 1e-10) -> bool:
    """Nonzero-overlap-add test on the host, in the window's own float32 like the reference
    (stft.py:382-431, itself scipy.signal.check_NOLA): fold w**2 onto one hop — whole hops first,
    then the ragged tail onto the leading positions — and require every folded position > tol."""
    w2 = get_window(window, n_fft, fftbins=True, device="cpu").numpy() ** 2
    whole, tail = divmod(n_fft, hop_length)
    folded = np.add.reduce(w2[: whole * hop_length].reshape(whole, hop_length), axis=0) if whole \
        else 0          # hop > n_fft: the reference fails on the same subscript (TypeError)
    if tail:
        folded[:tail] += w2[n_fft - tail:]
    return bool(np.min(folded) > tol)
