"""DCT, MFCC and delta features — same API as /root/reference/mlx_audio_primitives/mfcc.py.

mfcc = fused melspectrogram kernel -> dB kernels (global-max clip) -> DCT contraction
kernel applied straight to the (B, M, T) layout (no transposes, mfcc.py:265-271), with
the lifter folded into the DCT epilogue.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import _extension as _x
from ._validation import validate_positive
from .convert import power_to_db
from .mel import melspectrogram

_dct_cache: dict[tuple, torch.Tensor] = {}


def _dct_matrix(n_out: int, n_in: int, norm, device) -> torch.Tensor:
    """DCT-II basis (n_out, n_in): the float32 builder of the reference's native path
    (dct.cpp:24-101, used by mfcc.py:113-115 when its extension is present)."""
    key = (n_out, n_in, norm, str(device))
    t = _dct_cache.get(key)
    if t is None:
        t = torch.from_numpy(_x.dct_matrix_host(n_out, n_in, norm)).to(device)
        _dct_cache[key] = t
    return t


def dct(x, type: int = 2, n: int | None = None, axis: int = -1, norm: str | None = "ortho",
        _row_scale=None) -> torch.Tensor:
    """DCT-II along `axis` (reference mfcc.py:69-140)."""
    if type != 2:
        raise ValueError(f"Only DCT type 2 is supported, got {type}")
    x = _x.to_device_f32(x)
    if x.ndim == 0:
        raise ValueError("x must have at least one dimension")
    ax = axis % x.ndim
    n_in = x.shape[ax]
    if n is None:
        n = n_in
    outer = int(np.prod(x.shape[:ax], dtype=np.int64)) if ax > 0 else 1
    inner = int(np.prod(x.shape[ax + 1:], dtype=np.int64)) if ax + 1 < x.ndim else 1
    C = _dct_matrix(int(n), int(n_in), norm, x.device)
    out = torch.empty(x.shape[:ax] + (int(n),) + x.shape[ax + 1:], dtype=torch.float32,
                      device=x.device)
    if out.numel():
        _x.check(_x.dlib(x.device).ap_dct_f32(_x.ptr(x), _x.ptr(C),
                                     None if _row_scale is None else _x.ptr(_row_scale), outer,
                                     int(n_in), inner, int(n), _x.ptr(out), _x.stream_ptr(x.device)))
    return out


def mfcc(y=None, sr: int = 22050, S=None, n_mfcc: int = 20, dct_type: int = 2, norm: str | None = "ortho",
         lifter: int = 0, n_fft: int = 2048, hop_length: int = 512, win_length: int | None = None,
         window="hann", center: bool = True, pad_mode: str = "constant", power: float = 2.0,
         n_mels: int = 128, fmin: float = 0.0, fmax: float | None = None, htk: bool = False,
         mel_norm: str | None = "slaney") -> torch.Tensor:
    """Mel-frequency cepstral coefficients (reference mfcc.py:143-287).

    Returns (n_mfcc, n_frames) or (batch, n_mfcc, n_frames).  (A batch sharded by clip over several GPUs:
    ``sharding.mfcc_sharded``, which all-reduces the 4-byte global maximum the dB stage clips against.)"""
    return _mfcc_impl(y, sr, S, n_mfcc, dct_type, norm, lifter, n_fft, hop_length, win_length, window, center,
                      pad_mode, power, n_mels, fmin, fmax, htk, mel_norm, None)


def _mfcc_impl(y, sr, S, n_mfcc, dct_type, norm, lifter, n_fft, hop_length, win_length, window, center, pad_mode,
               power, n_mels, fmin, fmax, htk, mel_norm, max_reduce):
    """`mfcc`; ``max_reduce`` (None, or a callable on the 1-element int32 key tensor of max(S)) runs between the
    mel kernel and the dB + DCT kernel: the hook a clip-sharded batch needs (convert.py:58 clips against the
    maximum of the WHOLE batch)."""
    validate_positive(n_mfcc, "n_mfcc")
    provided = S is not None
    max_key = None
    if S is None:
        # the mel kernel also leaves max(S) for the top_db clip of the dB stage (one atomic per wave)
        from .mel import _is_pcm16, _melspectrogram_max, _to_device_pcm16
        y = _to_device_pcm16(y) if _is_pcm16(y) else _x.to_device_f32(y)
        max_key = torch.empty(1, dtype=torch.int32, device=y.device)
        S = _melspectrogram_max(y, sr, n_fft, hop_length, win_length, window, center, pad_mode, power, n_mels,
                                fmin, fmax, htk, mel_norm, max_key)
        if S.numel() == 0:
            max_key = None
        elif max_reduce is not None:
            max_reduce(max_key)
    else:
        S = _x.to_device_f32(S)
    batched = S.ndim == 3
    if not batched:
        S = S[None, :]
    lift = None
    if lifter > 0:
        nn = np.arange(n_mfcc)
        lift = torch.from_numpy(
            (1 + (lifter / 2.0) * np.sin(np.pi * (nn + 1) / lifter)).astype(np.float32)).to(S.device)
    if provided:
        # a provided S is taken to be log-power already (mfcc.py:253-258)
        M = dct(S, type=dct_type, n=n_mfcc, axis=1, norm=norm, _row_scale=lift)
    else:
        M = _db_dct(S, dct_type, n_mfcc, norm, lift, max_key)
    return M if batched else M[0]


def _db_dct(S: torch.Tensor, dct_type: int, n_mfcc: int, norm, lift, max_key=None) -> torch.Tensor:
    """power_to_db(S, ref=1.0, amin=1e-10, top_db=80.0) followed by the DCT along axis 1
    (mfcc.py:259-287) in one pass over S: ap_db_dct_f32 converts on load, so the dB array is
    never written.  Falls back to the two calls when the basis does not fit LDS."""
    if dct_type != 2:
        raise ValueError(f"Only DCT type 2 is supported, got {dct_type}")
    B, n_in, inner = S.shape
    if n_in * (16 if n_mfcc <= 16 else 32) * 4 > 64 * 1024:
        return dct(power_to_db(S, ref=1.0, amin=1e-10, top_db=80.0), type=dct_type, n=n_mfcc, axis=1,
                   norm=norm, _row_scale=lift)
    C = _dct_matrix(int(n_mfcc), int(n_in), norm, S.device)
    out = torch.empty((B, int(n_mfcc), inner), dtype=torch.float32, device=S.device)
    if out.numel():
        ws = max_key if max_key is not None else torch.empty(1, dtype=torch.int32, device=S.device)
        _x.check(_x.dlib(S.device).ap_db_dct_f32(_x.ptr(S), _x.ptr(C), None if lift is None else _x.ptr(lift),
                                        B, int(n_in), inner, int(n_mfcc), 10.0, 1e-10, 1.0, None, 80.0,
                                        ws.data_ptr(), int(max_key is not None), _x.ptr(out),
                                        _x.stream_ptr(S.device)))
    return out


# --------------------------------------------------------------------------- delta
_SG_MODES = {"interp": 0, "nearest": 1, "mirror": 2, "constant": 3, "wrap": 4}
_sg_cache: dict[tuple, tuple] = {}


def _savgol_tables(width: int, polyorder: int, deriv: int, delta_x: float, device):
    """Correlation-order Savitzky-Golay taps and the (2*half, width) edge rows of mode 'interp', built
    in float64 the way scipy.signal.savgol_coeffs / _fit_edge do (least-squares polynomial through
    `width` points, `deriv`-th derivative at the output position) - what the reference's host call
    scipy.signal.savgol_filter(data, width, deriv=order, polyorder=order) computes (mfcc.py:364-366)."""
    key = (width, polyorder, deriv, float(delta_x), str(device))
    hit = _sg_cache.get(key)
    if hit is not None:
        return hit
    if polyorder >= width:
        raise ValueError("polyorder must be less than window_length.")
    half = width // 2
    if deriv > polyorder:
        taps = np.zeros(width)
    else:
        pos = np.arange(-half, half + 1, dtype=np.float64)
        A = pos[None, :] ** np.arange(polyorder + 1)[:, None]               # (polyorder+1, width)
        rhs = np.zeros(polyorder + 1)
        rhs[deriv] = float(math.factorial(deriv)) / (delta_x ** deriv)
        taps = np.linalg.lstsq(A, rhs, rcond=None)[0]
    # edges: polynomial fitted to the first / last `width` samples, derivative evaluated at the positions
    # 0..half-1 (head) and width-half..width-1 (tail); linear in the samples -> one row per output
    coef = np.polyfit(np.arange(width, dtype=np.float64), np.eye(width), polyorder)   # (polyorder+1, width)
    rows = []
    for i in list(range(half)) + list(range(width - half, width)):
        row = np.empty(width)
        for j in range(width):
            pc = np.polyder(coef[:, j], deriv) if deriv > 0 else coef[:, j]
            row[j] = np.polyval(pc, float(i)) / (delta_x ** deriv)
        rows.append(row)
    edge = np.asarray(rows, dtype=np.float64)
    hit = (torch.from_numpy(taps.astype(np.float32)).to(device),
           torch.from_numpy(np.ascontiguousarray(edge, dtype=np.float32)).to(device))
    _sg_cache[key] = hit
    return hit


def delta(data, width: int = 9, order: int = 1, axis: int = -1, mode: str = "interp", **kwargs) -> torch.Tensor:
    """Delta (derivative) features by Savitzky-Golay filtering (reference mfcc.py:290-368, a
    scipy.signal.savgol_filter host call there; one FIR kernel along `axis` here).

    ``kwargs``: ``polyorder`` (default = order), ``delta`` (sample spacing), ``cval`` (mode 'constant')."""
    validate_positive(width, "width")
    validate_positive(order, "order")
    if width < 3:
        raise ValueError(f"width must be >= 3, got {width}")
    if width % 2 == 0:
        raise ValueError(f"width must be odd, got {width}")
    if mode not in _SG_MODES:
        raise ValueError("mode must be 'mirror', 'constant', 'nearest' 'wrap' or 'interp'.")
    x = _x.to_device_f32(data)
    if x.ndim == 0:
        x = x[None]
    ax = axis % x.ndim
    n = x.shape[ax]
    if mode == "interp" and width > n:
        raise ValueError(
            f"when mode='interp', width={width} "
            f"cannot exceed data.shape[axis]={n}"
        )
    kwargs.pop("deriv", None)
    polyorder = int(kwargs.pop("polyorder", order))
    delta_x = float(kwargs.pop("delta", 1.0))
    cval = float(kwargs.pop("cval", 0.0))
    if kwargs:
        raise TypeError(f"delta() got unexpected keyword arguments {sorted(kwargs)}")
    x = x.contiguous()
    taps, edge = _savgol_tables(int(width), polyorder, int(order), delta_x, x.device)
    outer = int(np.prod(x.shape[:ax], dtype=np.int64)) if ax > 0 else 1
    inner = int(np.prod(x.shape[ax + 1:], dtype=np.int64)) if ax + 1 < x.ndim else 1
    out = torch.empty_like(x)
    if out.numel():
        _x.check(_x.dlib(x.device).ap_savgol_f32(_x.ptr(x), outer, int(n), inner, _x.ptr(taps), int(width),
                                                 _SG_MODES[mode], cval, _x.ptr(edge), _x.ptr(out),
                                                 _x.stream_ptr(x.device)))
    return out
