"""Resampling — same API as /root/reference/mlx_audio_primitives/resample.py.

The reference ships every resampler to the host (SciPy / NumPy, resample.py:97,123,
279-281) and copies the result back.  Here ``resample_poly`` is a polyphase FIR kernel
that reproduces scipy.signal.resample_poly for float32 input (same taps, same
accumulation order), ``res_type='linear'`` is a float64-position interpolation kernel and
``res_type='fft'`` (scipy.signal.resample: one whole-clip FFT of arbitrary length) is a
four-step large-N FFT whose two legs both run in LDS (lengths N = N1*N2 with N1, N2 <= 4096;
any other length goes through the same engine as a chirp-z / Bluestein convolution).
"""

from __future__ import annotations

import math
from functools import lru_cache

import numpy as np
import torch

from . import _extension as _x
from ._validation import validate_positive

_taps_cache: dict[tuple, tuple] = {}


@lru_cache(maxsize=64)
def _poly_taps_host(up: int, down: int) -> tuple[bytes, int]:
    n = int(_x.lib().ap_resample_poly_ntaps(up, down))
    taps = np.empty(n, np.float32)
    import ctypes
    npr = ctypes.c_int(0)
    _x.check(_x.lib().ap_resample_poly_taps_host(up, down, taps.ctypes.data, ctypes.addressof(npr)))
    return taps.tobytes(), npr.value


def _poly_taps(up: int, down: int, device):
    key = (up, down, str(device))
    hit = _taps_cache.get(key)
    if hit is None:
        b, npr = _poly_taps_host(up, down)
        hit = (torch.from_numpy(np.frombuffer(b, dtype=np.float32).copy()).to(device), npr)
        _taps_cache[key] = hit
    return hit


def _as_rows(y, axis):
    """Move `axis` last and flatten the rest: (rows, L) contiguous float32 in HBM."""
    y = _x.to_device_f32(y)
    if y.ndim == 0:
        raise ValueError("y must have at least one dimension")
    ym = torch.movedim(y, axis, -1) if y.ndim > 1 else y
    lead = ym.shape[:-1]
    return ym.reshape(-1, ym.shape[-1]).contiguous(), lead


def _from_rows(out, lead, axis, ndim):
    out = out.reshape(*lead, out.shape[-1])
    return torch.movedim(out, -1, axis) if ndim > 1 else out


# scipy.signal.upfirdn's extension modes (AP_EXT_* in include/audioprims.h) and the padtypes SciPy
# implements by removing a per-row statistic first (scipy.signal.resample_poly: `funcs`)
_EXT_MODES = {"constant": 0, "wrap": 1, "edge": 2, "smooth": 3, "symmetric": 4, "reflect": 5,
              "antisymmetric": 6, "antireflect": 7, "line": 8}
_BACKGROUND = ("mean", "median", "minimum", "maximum")


def _row_background(rows: torch.Tensor, padtype: str) -> torch.Tensor:
    if padtype == "mean":
        return rows.mean(dim=-1, keepdim=True)
    if padtype == "minimum":
        return rows.amin(dim=-1, keepdim=True)
    if padtype == "maximum":
        return rows.amax(dim=-1, keepdim=True)
    srt = torch.sort(rows, dim=-1).values                # np.median: mean of the two middle values
    n = rows.shape[-1]
    return ((srt[:, (n - 1) // 2] + srt[:, n // 2]) / 2).unsqueeze(-1)


def resample_poly(y, up: int, down: int, axis: int = -1, padtype: str = "constant") -> torch.Tensor:
    """Polyphase resampling by up/down (reference resample.py:215-308 =
    scipy.signal.resample_poly(y, up, down, axis, padtype=padtype)).

    padtype: 'constant' (default), upfirdn's extension modes 'line', 'symmetric', 'reflect', 'edge',
    'wrap', 'smooth', 'antisymmetric', 'antireflect' (the signal is extended on the device and filtered
    by the same kernel), or 'mean' / 'median' / 'minimum' / 'maximum' (the row statistic is removed,
    the rest filtered against zeros, the statistic added back - as SciPy does)."""
    validate_positive(up, "up")
    validate_positive(down, "down")
    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    if up == 1 and down == 1:
        return y                                         # resample.py:259
    if padtype not in _EXT_MODES and padtype not in _BACKGROUND:
        raise ValueError(f"padtype must be one of {sorted(_BACKGROUND) + sorted(_EXT_MODES)}, got '{padtype}'")
    ndim = y.ndim if hasattr(y, "ndim") else np.ndim(y)
    rows, lead = _as_rows(y, axis)
    R, L = rows.shape
    n_out = (L * up + down - 1) // down
    out = torch.empty((R, n_out), dtype=torch.float32, device=rows.device)
    if R > 0 and L > 0:
        taps, n_pre_remove = _poly_taps(up, down, rows.device)
        background = None
        if padtype in _BACKGROUND:
            background = _row_background(rows, padtype)
            rows = rows - background
        mode = _EXT_MODES.get(padtype, 0)
        if mode == 0:
            _x.check(_x.dlib(rows.device).ap_resample_poly_f32(_x.ptr(rows), R, L, up, down, _x.ptr(taps),
                                                   taps.numel(), n_pre_remove, n_out, _x.ptr(out),
                                                   _x.stream_ptr(rows.device)))
        else:
            if L < 2 and padtype in ("smooth", "reflect", "antireflect", "line"):
                raise ValueError(f"padtype='{padtype}' needs at least two samples")
            n_ext = int(_x.lib().ap_resample_poly_pad_samples(up, down, taps.numel()))
            ws = torch.empty((R, L + 2 * n_ext), dtype=torch.float32, device=rows.device)
            _x.check(_x.dlib(rows.device).ap_resample_poly_padded_f32(
                _x.ptr(rows), R, L, up, down, _x.ptr(taps), taps.numel(), n_pre_remove, n_out, mode,
                _x.ptr(ws), _x.ptr(out), _x.stream_ptr(rows.device)))
        if background is not None:
            out += background
    return _from_rows(out, lead, axis, ndim)


def _cfft_split(n: int):
    """N = N1 * N2 with both legs LDS-resident (<= 4096), or None."""
    import ctypes
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    rc = _x.lib().ap_cfft_split_host(int(n), ctypes.addressof(a), ctypes.addressof(b))
    return (a.value, b.value) if rc == 0 else None


_chirp_cache: dict[tuple, tuple] = {}


def _chirp_tables(n: int, device):
    """Bluestein tables of a length the four-step engine cannot factor: chirp[k] = exp(-i pi k^2 / n)
    and the length-M spectrum of its conjugate laid out circularly, M = the power of two >= 2n - 1.
    Built once per length on the host in float64 (k^2 mod 2n keeps the phase exact), like the
    window and twiddle tables."""
    key = (n, str(device))
    hit = _chirp_cache.get(key)
    if hit is None:
        k = np.arange(n, dtype=np.int64)
        c = np.exp(-1j * np.pi * ((k * k) % (2 * n)) / n)
        m = 1 << int(2 * n - 2).bit_length()
        if _cfft_split(m) is None:
            raise ValueError(f"resample(res_type='fft'): length {n} is too long for the on-chip transform "
                             "(chirp length above 4096 * 4096); use resample_poly or res_type='linear'")
        b = np.zeros(m, np.complex128)
        b[:n] = np.conj(c)
        b[m - n + 1:] = np.conj(c[1:][::-1])
        spec = np.fft.fft(b)
        as_dev = lambda z: torch.from_numpy(np.ascontiguousarray(z.astype(np.complex64)).view(np.float32)).to(device)
        hit = (as_dev(c), as_dev(spec), m)
        _chirp_cache[key] = hit
    return hit


def _resample_fft(rows: torch.Tensor, out: torch.Tensor, post_scale: float) -> None:
    """scipy.signal.resample on the device (reference resample.py:97,123): complex FFT of the whole
    clip, SciPy's spectrum surgery, inverse transform.  Each transform runs on the four-step engine
    directly when its length factors as N1 * N2 with both <= 4096, otherwise as a chirp-z (Bluestein)
    convolution over the next power of two >= 2N - 1 on the same engine."""
    from .stft import _get_twiddles

    R, L = rows.shape
    n_out = out.shape[1]
    dev = rows.device
    args, nmax = [], max(L, n_out)
    for n in (L, n_out):
        split = _cfft_split(n)
        if split is not None:
            args += [0, _x.ptr(_get_twiddles(split[0], dev)), _x.ptr(_get_twiddles(split[1], dev)), None, None]
        else:
            chirp, spec, m = _chirp_tables(n, dev)
            m1, m2 = _cfft_split(m)
            args += [m, _x.ptr(_get_twiddles(m1, dev)), _x.ptr(_get_twiddles(m2, dev)), _x.ptr(chirp), _x.ptr(spec)]
            nmax = max(nmax, m)
    ws = torch.empty(4 * R * nmax, dtype=torch.float32, device=dev)
    _x.check(_x.dlib(dev).ap_resample_fft_chirp_f32(_x.ptr(rows), R, L, n_out, *args, _x.ptr(ws), _x.ptr(out),
                                                _x.stream_ptr(dev)))
    if post_scale != 1.0:
        out.mul_(post_scale)        # `scale=True`: y_new *= ratio (resample.py:126-127)


def _resample_fft_length(x: torch.Tensor, n_out: int) -> torch.Tensor:
    """scipy.signal.resample(x, n_out) along the last axis of a 1-D / 2-D float32 device tensor
    (the engine behind `_ext.resample_fft`, bindings.cpp:255-260)."""
    rows = x.reshape(-1, x.shape[-1]).contiguous()
    out = torch.empty((rows.shape[0], n_out), dtype=torch.float32, device=rows.device)
    if rows.shape[0] > 0:
        _resample_fft(rows, out, 1.0)
    return out.reshape(*x.shape[:-1], n_out)


def resample(y, orig_sr: int, target_sr: int, res_type: str = "fft", fix: bool = True,
             scale: bool = False, axis: int = -1) -> torch.Tensor:
    """Resample from orig_sr to target_sr (reference resample.py:21-212)."""
    validate_positive(orig_sr, "orig_sr")
    validate_positive(target_sr, "target_sr")
    if orig_sr == target_sr:
        return y
    if res_type not in ("fft", "linear"):
        raise ValueError(f"Unknown res_type: '{res_type}'. Supported: 'fft', 'linear'")
    ndim = y.ndim if hasattr(y, "ndim") else np.ndim(y)
    rows, lead = _as_rows(y, axis)
    R, L = rows.shape
    ratio = target_sr / orig_sr
    n_out = int(np.round(L * ratio)) if fix else int(np.ceil(L * ratio))
    if n_out == L:
        return _from_rows(rows, lead, axis, ndim)
    out = torch.empty((R, n_out), dtype=torch.float32, device=rows.device)
    if res_type == "fft":
        if R > 0 and n_out > 0:
            _resample_fft(rows, out, float(ratio) if scale else 1.0)
        return _from_rows(out, lead, axis, ndim)
    if R > 0 and n_out > 0:
        _x.check(_x.dlib(rows.device).ap_resample_linear_f32(_x.ptr(rows), R, L, n_out,
                                                 float(ratio) if scale else 1.0, _x.ptr(out),
                                                 _x.stream_ptr(rows.device)))
    return _from_rows(out, lead, axis, ndim)
