"""Griffin-Lim phase reconstruction — same API and the same update rule as
/root/reference/mlx_audio_primitives/griffinlim.py:17-196 (which is NOT librosa's
update; parity is with the reference).

Per iteration: one istft (irfft + overlap-add kernels), one stft (fused kernel) and ONE
element-wise projection kernel that does angle -> S*exp(i angle) -> momentum -> tprev
(the reference uses ~10 full-tensor ops for that, griffinlim.py:156-178).
The initial phase comes from host NumPy ``default_rng(random_state)`` over the full
(B,F,T) tensor, drawn in float64 and cast to float32, exactly as the reference does
(griffinlim.py:112-115), so seeds reproduce.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _extension as _x
from ._validation import validate_positive, validate_range
from .stft import istft, stft


_RNG_CHUNK = 1 << 20


_pinned: dict[str, torch.Tensor] = {}


def _staging(n: int, pinned: bool) -> torch.Tensor:
    """Reusable host staging buffer (page-locking 50+ MB per call costs more than the draw)."""
    if not pinned:
        return torch.empty(n, dtype=torch.float32)
    buf = _pinned.get("buf")
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 22), dtype=torch.float32).pin_memory()
        _pinned["buf"] = buf
    return buf[:n]


def _random_phase(random_state, shape, device) -> torch.Tensor:
    """uniform(-pi, pi, shape) from np.random.default_rng(random_state), float64 -> float32,
    bit-identical to the reference's single host call (griffinlim.py:112-115).

    PCG64 spends exactly one 64-bit output per double, so the stream is cut into chunks
    whose generators are advanced to their offset and filled by a small thread pool
    (NumPy releases the GIL while filling); each chunk is copied to the GPU asynchronously
    from pinned memory while the next ones are still being drawn."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    n = int(np.prod(shape))
    out = torch.empty(n, dtype=torch.float32, device=device)
    if n == 0:
        return out.reshape(shape)
    seed_bg = np.random.default_rng(random_state).bit_generator
    state = seed_bg.state
    n_chunks = (n + _RNG_CHUNK - 1) // _RNG_CHUNK
    host = _staging(n, torch.device(device).type == "cuda")
    host_np = host.numpy()

    def fill(c):
        lo = c * _RNG_CHUNK
        hi = min(n, lo + _RNG_CHUNK)
        bg = np.random.PCG64()
        bg.state = state
        bg.advance(lo)
        host_np[lo:hi] = np.random.Generator(bg).uniform(-np.pi, np.pi, hi - lo)   # f64 -> f32 cast
        return lo, hi

    workers = max(1, min(16, len(os.sched_getaffinity(0)), n_chunks))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        for lo, hi in pool.map(fill, range(n_chunks)):
            out[lo:hi].copy_(host[lo:hi], non_blocking=True)
    if out.is_cuda:
        torch.cuda.current_stream(out.device).synchronize()   # staging buffer is reused by the next call
    return out.reshape(shape)


def _project(mode, S, angles, R, momentum, tprev, rebuilt):
    B, F, T = S.shape
    TR = R.shape[-1] if R is not None else 0
    _x.check(_x.lib().ap_gl_project_f32(
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

    rebuilt = torch.empty((B, F, T), dtype=torch.complex64, device=dev)
    tprev = torch.empty((B, F, T), dtype=torch.complex64, device=dev)
    _project(0, S, angles, None, 0.0, tprev, rebuilt)          # rebuilt = tprev = S*exp(i*angles)
    del angles
    kw = dict(hop_length=hop_length, win_length=win_length, n_fft=n_fft, window=window,
              center=center)
    for _ in range(n_iter):
        y = istft(rebuilt, length=length, **kw)
        R = stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                 center=center, pad_mode=pad_mode)
        _project(1, S, None, R, momentum, tprev, rebuilt)
    y = istft(rebuilt, length=length, **kw)
    return y if batched else y[0]
