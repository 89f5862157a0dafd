"""Window functions — same API as /root/reference/mlx_audio_primitives/windows.py:192-256.

Windows are built on the host in float64 inside the C library
(``ap_generate_window_host``; reference windows.cpp:179-228, the path the reference
takes when its extension is present, windows.py:152-164), cached as bytes, and
cached again per device as tensors (the reference's two-tier cache,
windows.py:125-135).
"""

from __future__ import annotations

from functools import lru_cache

import numpy as np
import torch

from . import _extension as _x

_SUPPORTED = ("bartlett", "blackman", "boxcar", "hamming", "hann", "hanning", "ones",
              "rectangular", "triangular")

_device_window_cache: dict[tuple, torch.Tensor] = {}


@lru_cache(maxsize=128)
def _get_window_cached(window_name: str, n_fft: int, fftbins: bool) -> bytes:
    name = window_name.lower()
    if name not in _x.WINDOW_KINDS:
        raise ValueError(
            f"Unknown window type: '{name}'. Supported: {', '.join(_SUPPORTED)}"
        )
    return _x.generate_window_host(name, n_fft, fftbins).tobytes()


def _default_device(device=None) -> torch.device:
    if device is not None:
        return torch.device(device)
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def get_window(window, n_fft: int, fftbins: bool = True, device=None) -> torch.Tensor:
    """Window of shape (n_fft,), float32 (reference windows.py:192-256).

    ``window`` is a name or a 1-D tensor/array of length n_fft (passed through as
    float32).  Host-built; lands on the current HIP device when there is one."""
    if isinstance(window, (torch.Tensor, np.ndarray)):
        if window.shape[0] != n_fft:
            raise ValueError(
                f"Window array length ({window.shape[0]}) must match n_fft ({n_fft})"
            )
        t = torch.as_tensor(window)
        return t.to(device=_default_device(device) if device is not None else t.device,
                    dtype=torch.float32)
    if not isinstance(window, str):
        raise TypeError(f"window must be str or mx.array, got {type(window).__name__}")
    dev = _default_device(device)
    key = (window.lower(), n_fft, bool(fftbins), str(dev))
    hit = _device_window_cache.get(key)
    if hit is not None:
        return hit
    w = np.frombuffer(_get_window_cached(window, int(n_fft), bool(fftbins)), dtype=np.float32)
    t = torch.from_numpy(w.copy()).to(dev)
    _device_window_cache[key] = t
    return t
