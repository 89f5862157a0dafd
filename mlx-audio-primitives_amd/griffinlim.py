"""Griffin-Lim phase reconstruction — same API and the same update rule as
/root/reference/mlx_audio_primitives/griffinlim.py:17-196 (which is NOT librosa's
update; parity is with the reference).

Per iteration: one istft (irfft + overlap-add kernels), one stft (fused kernel) and ONE
element-wise projection kernel that does angle -> S*exp(i angle) -> momentum -> tprev
(the reference uses ~10 full-tensor ops for that, griffinlim.py:156-178); the whole loop is
enqueued by one C call (``ap_griffinlim_f32``).  The initial phase is NumPy's
``default_rng(random_state).uniform(-pi, pi, (B,F,T)).astype(float32)`` reproduced bit for bit
ON THE DEVICE (``ap_pcg64_uniform_f32``), so seeds reproduce the reference's draw
(griffinlim.py:112-115) without a 50 ms host RNG + PCIe copy.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _extension as _x
from ._validation import validate_positive, validate_range
from .stft import istft, magnitude, phase, stft


def _random_phase(random_state, shape, device) -> torch.Tensor:
    """uniform(-pi, pi, shape) of np.random.default_rng(random_state), float64 -> float32, drawn ON
    THE DEVICE and bit-identical to the reference's host call (griffinlim.py:112-115): the seed is
    expanded by NumPy on the host (SeedSequence -> PCG64 state, a few hundred bytes), the stream
    itself is produced by ap_pcg64_uniform_f32 (128-bit LCG jump-ahead + XSL-RR per element)."""
    n = int(np.prod(shape))
    dev = torch.device(device)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    if n:
        st = np.random.default_rng(random_state).bit_generator.state["state"]
        m = (1 << 64) - 1
        _x.check(_x.dlib(dev).ap_pcg64_uniform_f32(st["state"] >> 64, st["state"] & m, st["inc"] >> 64,
                                               st["inc"] & m, -np.pi, np.pi, n, _x.ptr(out),
                                               _x.stream_ptr(dev)))
    return out.reshape(shape)


def _project(mode, S, angles, R, momentum, tprev, rebuilt):
    B, F, T = S.shape
    TR = R.shape[-1] if R is not None else 0
    _x.check(_x.dlib(S.device).ap_gl_project_f32(
        mode, _x.ptr(S), None if angles is None else _x.ptr(angles),
        None if R is None else _x.ptr(torch.view_as_real(R)), TR, B * F, T, float(momentum),
        None if tprev is None else _x.ptr(torch.view_as_real(tprev)),
        _x.ptr(torch.view_as_real(rebuilt)), _x.stream_ptr(S.device)))


def griffinlim(S, n_iter: int = 32, hop_length: int | None = None, win_length: int | None = None,
               n_fft: int | None = None, window="hann", center: bool = True,
               length: int | None = None, pad_mode: str = "constant", momentum: float = 0.99,
               init: str = "random", random_state: int | None = None) -> torch.Tensor:
    """Reconstruct a waveform from a magnitude spectrogram S (F,T) or (B,F,T)."""
    validate_positive(n_iter, "n_iter")
    validate_range(momentum, "momentum", min_val=0.0, max_val=1.0, max_inclusive=False)
    S = _x.to_device_f32(S)
    batched = S.ndim == 3
    if not batched:
        S = S[None, :]
    S = S.contiguous()
    B, F, T = S.shape
    dev = S.device
    if n_fft is None:
        n_fft = 2 * (F - 1)
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft

    if init == "random":
        angles = _random_phase(random_state, (B, F, T), dev)
    elif init == "zeros":
        angles = torch.zeros((B, F, T), dtype=torch.float32, device=dev)
    else:
        raise ValueError(f"Unknown init: '{init}'. Supported: 'random', 'zeros'")

    # the whole loop is enqueued by ONE C call (no per-iteration Python / allocator work)
    from .stft import _frame_count, _get_padded_window, _get_twiddles
    if F != n_fft // 2 + 1:
        raise ValueError(f"S has {F} frequency bins but n_fft={n_fft} needs {n_fft // 2 + 1}")
    pad = n_fft // 2 if center else 0
    natural = n_fft + (T - 1) * hop_length
    y_len = length if length is not None else natural - 2 * pad
    if y_len <= 0:
        raise ValueError("griffinlim: the reconstructed signal would be empty")
    TR = _frame_count(y_len, n_fft, hop_length, center, pad_mode)
    win = _get_padded_window(window, win_length, n_fft, dev)
    tw = _get_twiddles(n_fft, dev)
    y = torch.empty((B, y_len), dtype=torch.float32, device=dev)
    if n_fft == 2048 and TR == T and _x.lib().ap_istft_workspace_floats(B, T, 2048, int(hop_length), pad) == 0:
        # n_fft = 2048: the three complex workspaces live with their rows padded to whole 128-byte lines
        # (16 frames), which the STFT kernel then writes and the ISTFT kernel reads without straddling lines
        Ts = -(-T // 16) * 16
        ws3 = torch.empty((3, B, F, Ts, 2), dtype=torch.float32, device=dev)
        rc = _x.dlib(dev).ap_griffinlim_rows_f32(
            _x.ptr(S), _x.ptr(angles), B, T, Ts, int(n_fft), int(hop_length), _x.ptr(win), _x.ptr(tw),
            int(bool(center)), _x.PAD_MODES[pad_mode], pad, y_len, int(n_iter), float(momentum),
            _x.ptr(ws3[0]), _x.ptr(ws3[1]), _x.ptr(ws3[2]), _x.ptr(y), _x.stream_ptr(dev))
        if rc != _x.AP_ERR_UNSUPPORTED:
            _x.check(rc)
            return y if batched else y[0]
    rebuilt = torch.empty((B, F, T, 2), dtype=torch.float32, device=dev)
    tprev = torch.empty((B, F, T, 2), dtype=torch.float32, device=dev)
    R = torch.empty((B, F, TR, 2), dtype=torch.float32, device=dev)
    n_ws = int(_x.lib().ap_istft_workspace_floats(B, T, int(n_fft), int(hop_length), pad))
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
    _x.check(_x.dlib(dev).ap_griffinlim_f32(
        _x.ptr(S), _x.ptr(angles), B, T, int(n_fft), int(hop_length), _x.ptr(win), _x.ptr(tw),
        int(bool(center)), _x.PAD_MODES[pad_mode], pad, y_len, TR, int(n_iter), float(momentum),
        _x.ptr(rebuilt), _x.ptr(tprev), _x.ptr(R), _x.ptr(ws), _x.ptr(y), _x.stream_ptr(dev)))
    return y if batched else y[0]


def griffinlim_iter(S, angles, hop_length: int, win_length: int, n_fft: int, window="hann", center: bool = True,
                    pad_mode: str = "constant", momentum: float = 0.99, tprev=None):
    """One Griffin-Lim iteration (reference griffinlim.py:199-284), for custom stopping criteria.

    Returns ``(new_angles, new_rebuilt, error)``: the phase of stft(istft(S exp(i angles))), the
    magnitude-constrained estimate S exp(i new_angles) with the momentum term ``+ momentum * (. - tprev)``
    when ``tprev`` is given, and the reconstruction error mean((S - |stft(istft(.))|)**2) as a 0-d tensor.
    Built from the library's own entry points: ap_gl_project_f32 (both projections), the fused istft / stft
    kernels, ap_magnitude / ap_phase and the deterministic ap_mse_f32 reduction."""
    S = _x.to_device_f32(S)
    batched = S.ndim == 3
    if S.ndim not in (2, 3):
        raise ValueError(f"S must be 2D or 3D, got {S.ndim}D")
    S3 = (S if batched else S[None]).contiguous()
    dev = S3.device
    ang = _x.to_device_f32(angles, dev)
    ang3 = (ang if ang.ndim == 3 else ang[None]).contiguous()
    if ang3.shape != S3.shape:
        raise ValueError(f"angles must have the shape of S, got {tuple(ang.shape)} and {tuple(S.shape)}")
    B, F, T = S3.shape
    rebuilt = torch.empty((B, F, T), dtype=torch.complex64, device=dev)
    _project(0, S3, ang3, None, 0.0, None, rebuilt)                     # S exp(i angles)
    y_est = istft(rebuilt, hop_length=hop_length, win_length=win_length, n_fft=n_fft, window=window, center=center)
    R = stft(y_est, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, center=center,
             pad_mode=pad_mode)
    if R.shape != S3.shape:            # the reference's S - mag_new fails to broadcast on the same mismatch
        raise ValueError(f"stft of the estimate has shape {tuple(R.shape)}, S has {tuple(S3.shape)}")
    mag_new = magnitude(R)
    new_angles = phase(R)
    ws = torch.empty(int(_x.lib().ap_mse_workspace_doubles()), dtype=torch.float64, device=dev)
    err = torch.empty(1, dtype=torch.float32, device=dev)
    _x.check(_x.dlib(dev).ap_mse_f32(_x.ptr(S3), _x.ptr(mag_new), S3.numel(), _x.ptr(ws), _x.ptr(err),
                                     _x.stream_ptr(dev)))
    Rd = R.contiguous()
    out = torch.empty((B, F, T), dtype=torch.complex64, device=dev)
    if momentum > 0 and tprev is not None:
        tp = tprev if isinstance(tprev, torch.Tensor) else torch.as_tensor(np.asarray(tprev))
        tp = tp.to(device=dev, dtype=torch.complex64)
        tp = (tp if tp.ndim == 3 else tp[None]).clone().contiguous()    # the projection kernel updates its tprev in place
        if tp.shape != S3.shape:
            raise ValueError(f"tprev must have the shape of S, got {tuple(tp.shape)}")
        _project(1, S3, None, Rd, momentum, tp, out)
    else:
        _project(1, S3, None, Rd, 0.0, None, out)
    if not batched:
        new_angles, out = new_angles[0], out[0]
    return new_angles, out, err[0]
