"""Time-domain primitives — same API as /root/reference/mlx_audio_primitives/framing.py
(frame / rms / preemphasis / deemphasis).

rms (and features.zero_crossing_rate) never materialise the (B, T, frame_length) frame tensor the
reference builds (framing.py:135-139): one kernel stages each clip span in LDS and reduces the frames
from there.  deemphasis, a SciPy lfilter call on the host in the reference (framing.py:354-380), is a
blocked scan of the first-order recursion on the device.
"""

from __future__ import annotations

import torch

from . import _extension as _x
from ._validation import validate_positive


def frame(y, frame_length: int, hop_length: int, axis: int = -1) -> torch.Tensor:
    """Overlapping frames (n_frames, frame_length) / (batch, n_frames, frame_length)
    (reference framing.py:16-78)."""
    validate_positive(frame_length, "frame_length")
    validate_positive(hop_length, "hop_length")
    if axis != -1:
        raise ValueError(f"axis must be -1, got {axis}")
    _x.lib()                              # loud failure when the HIP extension is missing
    return _x._ext.frame_signal(y, int(frame_length), int(hop_length))


def _frame_stats(y, frame_length, hop_length, center, pad_mode, which):
    validate_positive(frame_length, "frame_length")
    validate_positive(hop_length, "hop_length")
    y = _x.to_device_f32(y)
    one_d = y.ndim == 1
    if one_d:
        y = y[None, :]
    if y.ndim != 2:
        raise ValueError(f"y must be 1D or 2D, got {y.ndim}D")
    if center and pad_mode not in ("constant", "edge"):
        raise ValueError(f"Unknown pad_mode: '{pad_mode}'. Supported: 'constant', 'edge'")
    B, L = y.shape
    pad = frame_length // 2 if center else 0
    Lp = L + 2 * pad
    if Lp < frame_length:
        raise ValueError(
            f"Signal length ({Lp}) must be >= frame_length ({frame_length}). "
            f"Consider padding the signal."
        )
    T = 1 + (Lp - frame_length) // hop_length
    out = torch.empty((B, 1, T), dtype=torch.float32, device=y.device)
    if B > 0:
        _x.check(_x.dlib(y.device).ap_frame_stats_f32(
            _x.ptr(y), B, L, int(frame_length), int(hop_length), int(bool(center)),
            _x.PAD_MODES[pad_mode if center else "constant"], T,
            _x.ptr(out) if which == "rms" else None, _x.ptr(out) if which == "zcr" else None,
            _x.stream_ptr(y.device)))
    return out[0] if one_d else out


def rms(y, frame_length: int = 2048, hop_length: int = 512, center: bool = True,
        pad_mode: str = "constant") -> torch.Tensor:
    """Root-mean-square energy per frame, (1, T) / (batch, 1, T) (reference framing.py:81-150)."""
    return _frame_stats(y, frame_length, hop_length, center, pad_mode, "rms")


def _row_state(zi, B, device):
    """Initial filter state as a (B,) device tensor (scalar / (B,) / (B,1) accepted,
    reference framing.py:238-250,356-365)."""
    if zi is None:
        return None
    z = _x.to_device_f32(zi, device).reshape(-1)
    if z.numel() == 1:
        z = z.expand(B)
    if z.numel() != B:
        raise ValueError(f"zi must be a scalar or have one entry per signal ({B}), got {z.numel()}")
    return z.contiguous()


def _emphasis(fn_name, y, coef, zi, return_zf):
    if not 0.0 <= coef <= 1.0:
        raise ValueError(f"coef must be in [0, 1], got {coef}")
    y = _x.to_device_f32(y)
    one_d = y.ndim == 1
    if one_d:
        y = y[None, :]
    if y.ndim != 2:
        raise ValueError(f"y must be 1D or 2D, got {y.ndim}D")
    B, L = y.shape
    dev = y.device
    z = _row_state(zi, B, dev)
    out = torch.empty_like(y)
    zf = torch.empty((B, 1), dtype=torch.float32, device=dev)
    if B > 0 and L > 0:
        if fn_name == "ap_deemphasis_f32":
            # long clips: chunk end states in a small workspace, every chunk on a workgroup of its own
            n_ws = int(_x.lib().ap_deemphasis_workspace_floats(B, L))
            ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
            _x.check(_x.dlib(dev).ap_deemphasis_ws_f32(_x.ptr(y), B, L, float(coef), None if z is None else _x.ptr(z),
                                                       _x.ptr(out), _x.ptr(zf), _x.ptr(ws) if n_ws else None,
                                                       _x.stream_ptr(dev)))
        else:
            _x.check(getattr(_x.dlib(dev), fn_name)(_x.ptr(y), B, L, float(coef), None if z is None else _x.ptr(z),
                                                     _x.ptr(out), _x.ptr(zf), _x.stream_ptr(dev)))
    if one_d:
        out, zf = out[0], zf[0]
    return (out, zf) if return_zf else out


def preemphasis(y, coef: float = 0.97, zi=None, return_zf: bool = False, use_mlx: bool = True):
    """y[n] - coef*y[n-1]; first sample y[0] + zi, zi = 2*y[0] - y[1] by default
    (reference framing.py:194-296; ``use_mlx`` accepted for signature compatibility)."""
    return _emphasis("ap_preemphasis_f32", y, coef, zi, return_zf)


def deemphasis(y, coef: float = 0.97, zi=None, return_zf: bool = False):
    """Inverse of preemphasis: out[n] = y[n] + coef*out[n-1] (reference framing.py:298-392)."""
    return _emphasis("ap_deemphasis_f32", y, coef, zi, return_zf)
