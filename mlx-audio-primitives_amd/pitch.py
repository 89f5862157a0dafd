"""Autocorrelation — same API as /root/reference/mlx_audio_primitives/pitch.py:16-115 (native mirror
csrc/primitives/autocorrelation.cpp:10-84).

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
