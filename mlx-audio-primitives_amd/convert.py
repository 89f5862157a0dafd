"""power/amplitude <-> dB — same API as /root/reference/mlx_audio_primitives/convert.py.

``top_db`` clips against the GLOBAL maximum of the whole (batched) array exactly like
the reference (convert.py:58): a two-kernel pass with one integer atomic per workgroup.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _extension as _x


def _to_db(S, ref, coefficient: float, amin: float, top_db):
    if top_db is not None and top_db <= 0:
        raise ValueError(f"top_db must be positive, got {top_db}")
    S = _x.to_device_f32(S)
    dev = S.device
    out = torch.empty_like(S)
    n = S.numel()
    if n == 0:
        return out
    ws = torch.empty(2, dtype=torch.int32, device=dev)       # [0] clip max key, [1] ref key
    ref_key = None
    ref_value = 0.0
    if callable(ref):
        if ref in (torch.max, np.max, max) or getattr(ref, "__name__", "") in ("max", "amax"):
            _x.check(_x.dlib(dev).ap_reduce_max_f32(_x.ptr(S), n, ws.data_ptr() + 4, _x.stream_ptr(dev)))
            ref_key = ws.data_ptr() + 4
        else:
            ref_value = float(ref(S))                         # arbitrary callable: host scalar
    else:
        ref_value = float(ref)
    _x.check(_x.dlib(dev).ap_to_db_f32(_x.ptr(S), n, float(coefficient), float(amin), ref_value, ref_key,
                                   -1.0 if top_db is None else float(top_db), _x.ptr(out),
                                   ws.data_ptr(), _x.stream_ptr(dev)))
    return out


def power_to_db(S, ref=1.0, amin: float = 1e-10, top_db: float | None = 80.0) -> torch.Tensor:
    """10*log10(S/ref) (reference convert.py:63-97)."""
    return _to_db(S, ref, 10.0, amin, top_db)


def amplitude_to_db(S, ref=1.0, amin: float = 1e-5, top_db: float | None = 80.0) -> torch.Tensor:
    """20*log10(S/ref) (reference convert.py:132-166)."""
    return _to_db(S, ref, 20.0, amin, top_db)


def _from_db(S_db, ref: float, div: float):
    S_db = _x.to_device_f32(S_db)
    out = torch.empty_like(S_db)
    n = S_db.numel()
    if n:
        _x.check(_x.dlib(S_db.device).ap_from_db_f32(_x.ptr(S_db), n, float(ref), div, _x.ptr(out),
                                         _x.stream_ptr(S_db.device)))
    return out


def db_to_power(S_db, ref: float = 1.0) -> torch.Tensor:
    """ref * 10**(S_db/10) (reference convert.py:100-129)."""
    return _from_db(S_db, ref, 10.0)


def db_to_amplitude(S_db, ref: float = 1.0) -> torch.Tensor:
    """ref * 10**(S_db/20) (reference convert.py:169-198)."""
    return _from_db(S_db, ref, 20.0)
