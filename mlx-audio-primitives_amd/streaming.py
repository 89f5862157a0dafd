"""Chunked (streaming) STFT / mel front end (the reference only lists
"Streaming support - process audio in chunks" as future work, ARCHITECTURE.md:537-540).

A stream is framed WITHOUT centring: frame t covers samples [t*hop, t*hop + n_fft) of the
concatenation of every chunk fed so far.  Each ``process(chunk)`` call returns exactly the frames the
new samples complete, computed by the same fused kernels as the offline calls, so concatenating the
outputs over any chunking equals ``stft(whole, center=False)`` / ``melspectrogram(whole,
center=False)`` bit for bit; only the n_fft - hop (or fewer) samples that later frames still need stay
in HBM between calls.  ``center=True`` semantics are obtained by feeding n_fft//2 zeros first and
calling ``flush()`` (which pads n_fft//2 zeros) at the end.
"""

from __future__ import annotations

import torch

from . import _extension as _x
from .mel import melspectrogram
from .stft import _resolve_stft_args, stft


class StreamingSTFT:
    """Incremental ``stft(..., center=False)``: feed (samples,) or (batch, samples) chunks."""

    def __init__(self, n_fft: int = 2048, hop_length: int | None = None, win_length: int | None = None,
                 window="hann", center: bool = False, **mel_kwargs):
        self.n_fft = int(n_fft)
        self.hop_length, self.win_length = _resolve_stft_args(self.n_fft, hop_length, win_length)
        self.window = window
        self.center = bool(center)
        self._mel = dict(mel_kwargs) if mel_kwargs else None      # n_mels=..., sr=..., power=... -> mel frames
        self._tail = None                                          # (B, < n_fft) samples not yet consumed
        self._one_d = None
        self._started = False
        self.frames_emitted = 0

    # -- internals ---------------------------------------------------------------------------------
    def _transform(self, y):
        if self._mel is not None:
            return melspectrogram(y, n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                                  window=self.window, center=False, **self._mel)
        return stft(y, n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                    window=self.window, center=False)

    def _empty(self, B, device):
        if self._mel is not None:
            n_rows = int(self._mel.get("n_mels", 128))
            return torch.empty((B, n_rows, 0), dtype=torch.float32, device=device)
        return torch.empty((B, self.n_fft // 2 + 1, 0), dtype=torch.complex64, device=device)

    def process(self, chunk) -> torch.Tensor:
        """Feed the next samples; returns the newly completed frames, (F|M, T_new) or (B, F|M, T_new)
        (T_new may be 0)."""
        chunk = _x.to_device_f32(chunk)
        if self._one_d is None:
            self._one_d = chunk.ndim == 1
        if chunk.ndim == 1:
            chunk = chunk[None, :]
        if chunk.ndim != 2:
            raise ValueError(f"chunk must be 1D or 2D, got {chunk.ndim}D")
        if not self._started and self.center:                      # the left half of the centre padding
            chunk = torch.nn.functional.pad(chunk, (self.n_fft // 2, 0))
        self._started = True
        buf = chunk if self._tail is None else torch.cat([self._tail, chunk], dim=1)
        if self._tail is not None and buf.shape[0] != self._tail.shape[0]:
            raise ValueError("every chunk must have the same batch size")
        n = buf.shape[1]
        T = 0 if n < self.n_fft else 1 + (n - self.n_fft) // self.hop_length
        if T == 0:
            self._tail = buf.contiguous()
            out = self._empty(buf.shape[0], buf.device)
        else:
            used = (T - 1) * self.hop_length + self.n_fft
            out = self._transform(buf[:, :used].contiguous())
            self._tail = buf[:, T * self.hop_length:].contiguous()
            self.frames_emitted += T
        return out[0] if self._one_d else out

    def flush(self) -> torch.Tensor:
        """End of stream.  center=True: pad the right half (n_fft//2 zeros) and emit the last frames;
        center=False: nothing is pending (a partial frame is dropped, as offline)."""
        if self._tail is None:
            raise ValueError("flush() before any chunk")
        if not self.center:
            return self._empty(self._tail.shape[0], self._tail.device)[0] if self._one_d else \
                self._empty(self._tail.shape[0], self._tail.device)
        pad = torch.zeros((self._tail.shape[0], self.n_fft // 2), dtype=torch.float32, device=self._tail.device)
        one_d, self._one_d = self._one_d, False
        self.center = False                                        # the padding is explicit from here on
        out = self.process(pad)
        self._one_d, self.center = one_d, True
        return out[0] if one_d else out

    def reset(self) -> None:
        self._tail, self._one_d, self._started, self.frames_emitted = None, None, False, 0
