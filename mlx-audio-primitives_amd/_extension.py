"""The drop-in boundary: loads libaudioprims_hip.so (C ABI, include/audioprims.h) and
publishes ``HAS_HIP_EXT`` / ``_ext`` the way the reference publishes
``HAS_CPP_EXT`` / ``_ext`` (/root/reference/mlx_audio_primitives/_extension.py:25-44).

Differences from the reference, on purpose:
  * there is NO framework fallback.  The reference silently degrades to an MLX op
    graph when its extension is missing; here every device op raises
    ``RuntimeError`` if the HIP library or a GPU is missing, so a silent slow path
    can never be mistaken for the product.
  * ``_ext`` is a thin Python object over ctypes whose methods have the reference
    nanobind signatures (csrc/bindings.cpp:15-368) but take/return torch tensors
    living in HBM.  PyTorch only supplies device memory and the current stream.
"""

from __future__ import annotations

import ctypes
import os
from typing import Any

import numpy as np

# IMPORTANT: import torch BEFORE dlopen()ing the extension.  The PyTorch-ROCm wheel
# bundles its own libamdhip64 (soname libamdhip64.so.7); loaded first, the dynamic
# linker reuses it for our library's NEEDED libamdhip64.so.7, so kernels, streams and
# allocations all live in ONE HIP runtime.  Loaded the other way round the process
# ends up with two runtimes and the first launch fails with "no ROCm-capable device".
# (The reference has the same ordering rule for mlx.core, _extension.py:22-25.)
import torch  # noqa: F401

from . import _build

_c_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int

PAD_MODES = {"constant": 0, "edge": 1, "reflect": 2}
WINDOW_KINDS = {
    "hann": 0, "hanning": 0, "hamming": 1, "blackman": 2, "bartlett": 3, "triangular": 3,
    "rectangular": 4, "boxcar": 4, "ones": 4,
}

#: every symbol include/audioprims.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "ap_version", "ap_last_error",
    "ap_generate_window_host", "ap_hz_to_mel_host", "ap_mel_to_hz_host",
    "ap_mel_filterbank_host", "ap_dct_matrix_host", "ap_twiddle_table_host", "ap_fft_supported",
    "ap_mel_plan_words", "ap_mel_plan_host",
    "ap_pad_f32", "ap_frame_f32", "ap_overlap_add_f32",
    "ap_stft_f32", "ap_stft_rows_f32", "ap_melspec_f32", "ap_melspec_max_f32", "ap_melspec_rows_fused", "ap_melspec_rows_f32", "ap_irfft_frames_f32", "ap_istft_f32", "ap_istft_rows_f32", "ap_istft_workspace_floats",
    "ap_magnitude_f32", "ap_phase_f32", "ap_complex_unary_rows_f32",
    "ap_resample_poly_ntaps", "ap_resample_poly_taps_host", "ap_resample_poly_f32",
    "ap_extend_f32", "ap_resample_poly_pad_samples", "ap_resample_poly_padded_f32", "ap_resample_fft_chirp_f32",
    "ap_resample_linear_f32", "ap_gl_project_f32", "ap_reduce_max_f32", "ap_to_db_f32",
    "ap_from_db_f32", "ap_dct_f32", "ap_db_dct_f32", "ap_cfft_split_host", "ap_resample_fft_f32",
    "ap_mse_workspace_doubles", "ap_mse_f32", "ap_pcg64_uniform_f32", "ap_griffinlim_f32", "ap_griffinlim_rows_f32",
    "ap_spectral_stats_f32", "ap_spectral_audio_fused", "ap_spectral_audio_f32", "ap_spectral_contrast_f32", "ap_frame_stats_f32", "ap_preemphasis_f32", "ap_deemphasis_f32", "ap_deemphasis_workspace_floats", "ap_deemphasis_ws_f32", "ap_savgol_f32",
    "ap_autocorrelation_nfft", "ap_autocorrelation_f32", "ap_acf_peaks_f32",
    "ap_pcm16_to_f32", "ap_melspec_pcm16_fused", "ap_melspec_pcm16_f32",
]

HAS_HIP_EXT: bool = False
_lib: Any | None = None
_load_error: str | None = None


def _declare(lib) -> None:
    P, I, L, F = ctypes.c_void_p, _int, _i64, ctypes.c_float
    lib.ap_version.restype = I
    lib.ap_last_error.restype = ctypes.c_char_p
    sig = {
        "ap_generate_window_host": [I, I, I, P],
        "ap_hz_to_mel_host": [P, L, I, P],
        "ap_mel_to_hz_host": [P, L, I, P],
        "ap_mel_filterbank_host": [I, I, I, ctypes.c_double, ctypes.c_double, I, I, P],
        "ap_dct_matrix_host": [I, I, I, P],
        "ap_twiddle_table_host": [I, P],
        "ap_fft_supported": [I],
        "ap_pad_f32": [P, L, L, L, I, P, P],
        "ap_frame_f32": [P, L, L, I, I, P, P],
        "ap_overlap_add_f32": [P, P, L, L, I, I, L, L, P, P],
        "ap_stft_f32": [P, L, L, I, I, P, P, I, I, L, P, P],
        "ap_stft_rows_f32": [P, L, L, I, I, P, P, I, I, L, L, P, P],
        "ap_melspec_f32": [P, L, L, I, I, P, P, I, I, L, P, P, P, I, F, P, P],
        "ap_melspec_max_f32": [P, L, L, I, I, P, P, I, I, L, P, P, P, I, F, P, P, P],
        "ap_melspec_rows_fused": [I, I, I, I, I, F, P, P],
        "ap_melspec_rows_f32": [P, L, L, I, I, P, P, I, I, L, L, P, P, P, I, F, P, P, P],
        "ap_mel_plan_host": [P, I, I, P, P],
        "ap_irfft_frames_f32": [P, L, L, I, P, P, P],
        "ap_istft_f32": [P, L, L, I, I, P, P, P, L, L, P, P],
        "ap_istft_rows_f32": [P, L, L, L, I, I, P, P, L, L, P, P],
        "ap_magnitude_f32": [P, L, P, P],
        "ap_complex_unary_rows_f32": [P, L, L, L, I, P, P],
        "ap_resample_poly_ntaps": [I, I],
        "ap_resample_poly_taps_host": [I, I, P, P],
        "ap_resample_poly_f32": [P, L, L, I, I, P, I, I, L, P, P],
        "ap_extend_f32": [P, L, L, L, I, P, P],
        "ap_resample_poly_pad_samples": [I, I, I],
        "ap_resample_poly_padded_f32": [P, L, L, I, I, P, I, I, L, I, P, P, P],
        "ap_resample_linear_f32": [P, L, L, L, ctypes.c_double, P, P],
        "ap_gl_project_f32": [I, P, P, P, L, L, L, F, P, P, P],
        "ap_reduce_max_f32": [P, L, P, P],
        "ap_to_db_f32": [P, L, F, F, F, P, F, P, P, P],
        "ap_from_db_f32": [P, L, F, F, P, P],
        "ap_dct_f32": [P, P, P, L, I, L, I, P, P],
        "ap_db_dct_f32": [P, P, P, L, I, L, I, F, F, F, P, F, P, I, P, P],
        "ap_cfft_split_host": [L, P, P],
        "ap_pcg64_uniform_f32": [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                 ctypes.c_double, ctypes.c_double, L, P, P],
        "ap_griffinlim_f32": [P, P, L, L, I, I, P, P, I, I, L, L, L, I, F, P, P, P, P, P, P],
        "ap_mse_f32": [P, P, L, P, P, P],
        "ap_griffinlim_rows_f32": [P, P, L, L, L, I, I, P, P, I, I, L, L, I, F, P, P, P, P, P],
        "ap_resample_fft_f32": [P, L, L, L, P, P, P, P, P, P, P],
        "ap_resample_fft_chirp_f32": [P, L, L, L, L, P, P, P, P, L, P, P, P, P, P, P, P],
        "ap_phase_f32": [P, L, P, P],
        "ap_spectral_stats_f32": [P, I, L, L, L, P, F, P, F, I, F, F, P, P, P, P, P],
        "ap_spectral_audio_fused": [L, I, I, I, I],
        "ap_spectral_contrast_f32": [P, L, L, L, P, I, I, P, P],
        "ap_spectral_audio_f32": [P, L, L, I, I, P, P, I, I, L, P, F, F, I, F, F, P, P, P, P, P],
        "ap_frame_stats_f32": [P, L, L, I, I, I, I, L, P, P, P],
        "ap_preemphasis_f32": [P, L, L, F, P, P, P, P],
        "ap_deemphasis_f32": [P, L, L, F, P, P, P, P],
        "ap_deemphasis_ws_f32": [P, L, L, F, P, P, P, P, P],
        "ap_savgol_f32": [P, L, L, L, P, I, I, F, P, P, P],
        "ap_autocorrelation_f32": [P, L, L, L, I, I, P, P, P, P, P],
        "ap_acf_peaks_f32": [P, L, I, I, I, F, F, P, P, P, P],
        "ap_pcm16_to_f32": [P, L, F, P, P],
        "ap_melspec_pcm16_fused": [L, I, I, I, I, I, F, P],
        "ap_melspec_pcm16_f32": [P, L, L, I, I, P, P, I, I, L, P, P, P, I, F, P, P, P, P],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = I
    lib.ap_mel_plan_words.argtypes = [P, I, I]
    lib.ap_mel_plan_words.restype = L
    lib.ap_istft_workspace_floats.argtypes = [L, L, I, I, L]
    lib.ap_istft_workspace_floats.restype = L
    lib.ap_autocorrelation_nfft.argtypes = [L]
    lib.ap_autocorrelation_nfft.restype = L
    lib.ap_resample_poly_pad_samples.restype = L
    lib.ap_deemphasis_workspace_floats.argtypes = [L, L]
    lib.ap_deemphasis_workspace_floats.restype = L
    lib.ap_mse_workspace_doubles.argtypes = []
    lib.ap_mse_workspace_doubles.restype = L


def _load() -> None:
    global HAS_HIP_EXT, _lib, _load_error
    path = _build.LIB_PATH
    try:
        if not os.path.exists(path):
            raise OSError(f"{path} not built (run __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        _declare(lib)
        # smoke-test one host entry, as the reference does with generate_window("hann", 4, True)
        buf = (ctypes.c_float * 4)()
        if lib.ap_generate_window_host(0, 4, 1, ctypes.cast(buf, ctypes.c_void_p)) != 0:
            raise OSError("ap_generate_window_host smoke test failed")
        _lib = lib
        HAS_HIP_EXT = True
        _load_error = None
    except (OSError, AttributeError) as e:
        _lib = None
        HAS_HIP_EXT = False
        _load_error = str(e)


_load()


def lib():
    """The raw ctypes handle; raises loudly when the HIP extension is missing."""
    if _lib is None:
        raise RuntimeError(
            "libaudioprims_hip.so is not available: "
            f"{_load_error}. There is no CPU/framework fallback on purpose."
        )
    return _lib


class _DeviceLib:
    """The library with `device` made the current HIP device around every call.

    hipLaunchKernelGGL, hipFuncSetAttribute (the > 48 KiB LDS opt-in) and hipMemsetD32Async act on
    the process's CURRENT device, while the buffers and the stream handed over belong to the
    tensors' device: without this guard a tensor on cuda:1 while cuda:0 is current would be
    launched on the wrong device with a foreign stream handle."""

    __slots__ = ("_index",)

    def __init__(self, index: int):
        self._index = index

    def __getattr__(self, name):
        fn = getattr(lib(), name)
        index = self._index

        def call(*args):
            import torch

            if torch.cuda.current_device() == index:
                return fn(*args)
            with torch.cuda.device(index):
                return fn(*args)

        return call


def dlib(device) -> _DeviceLib:
    """`lib()` bound to `device` (a torch.device / index): every entry point that enqueues
    device work is called through this."""
    import torch

    device = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
    if device.type != "cuda":
        raise RuntimeError(f"expected a HIP ('cuda') device, got {device}")
    index = device.index if device.index is not None else torch.cuda.current_device()
    return _DeviceLib(index)


AP_OK, AP_ERR_INVALID, AP_ERR_UNSUPPORTED, AP_ERR_HIP = 0, -1, -2, -3     # include/audioprims.h:37-39


def check(rc: int) -> None:
    """Turn a C status into the exception the reference would raise."""
    if rc == 0:
        return
    msg = lib().ap_last_error().decode()
    if rc in (-1, -2):
        raise ValueError(msg)       # std::invalid_argument -> ValueError in the reference
    raise RuntimeError(msg)


def require_device(device=None):
    """Resolve the HIP device to run on; no GPU is a hard error."""
    import torch

    lib()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "no HIP device visible: the audio primitives run only on a GPU "
            "(there is no CPU fallback)."
        )
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"expected a HIP ('cuda') device, got {device}")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def to_device_f32(x, device=None):
    """Contiguous float32 tensor in HBM (mirrors astype(float32)+contiguous at
    overlap_add.cpp:27-34)."""
    import torch

    if isinstance(x, torch.Tensor):
        if x.is_cuda and device is None:
            dev = x.device
        else:
            dev = require_device(device)
        lib()
        return x.to(device=dev, dtype=torch.float32).contiguous()
    dev = require_device(device)
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev).contiguous()


def stream_ptr(device) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream


def ptr(t) -> int:
    return t.data_ptr()


# --------------------------------------------------------------------------
# host builders (work without a GPU: pure host code inside the .so)
# --------------------------------------------------------------------------
def generate_window_host(window_type: str, length: int, periodic: bool = True) -> np.ndarray:
    kind = WINDOW_KINDS.get(window_type)
    if kind is None:
        raise ValueError(
            f"Unknown window type: '{window_type}'. "
            "Supported: hann, hamming, blackman, bartlett, rectangular"
        )
    if length <= 0:
        raise ValueError("Window length must be positive")
    out = np.empty(length, np.float32)
    check(lib().ap_generate_window_host(kind, int(length), int(bool(periodic)), out.ctypes.data))
    return out


def twiddle_table_host(n_fft: int) -> np.ndarray:
    out = np.empty(2 * n_fft, np.float32)
    check(lib().ap_twiddle_table_host(int(n_fft), out.ctypes.data))
    return out


def mel_filterbank_host(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False,
                        norm="slaney") -> np.ndarray:
    if norm not in ("slaney", "", None):
        raise ValueError(f"Unknown norm: '{norm}'. Supported: 'slaney', empty string for none")
    out = np.empty((max(int(n_mels), 0), 1 + max(int(n_fft), 0) // 2), np.float32)
    check(lib().ap_mel_filterbank_host(int(sr), int(n_fft), int(n_mels), float(fmin),
                                       -1.0 if fmax is None else float(fmax), int(bool(htk)),
                                       1 if norm == "slaney" else 0, out.ctypes.data))
    return out


PLAN_BANDED, PLAN_PARTS, PLAN_FORCE_GENERIC = 1, 2, 256


def mel_plan_host(fb: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Contraction plan of a dense (M, F) filterbank (include/audioprims.h:
    ap_mel_plan_host): (device-bound int32 blob, 16-int host descriptor)."""
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    M, F = fb.shape
    plan = np.zeros(int(lib().ap_mel_plan_words(fb.ctypes.data, M, F)), np.int32)
    desc = np.zeros(16, np.int32)
    check(lib().ap_mel_plan_host(fb.ctypes.data, M, F, plan.ctypes.data, desc.ctypes.data))
    return plan, desc


def dct_matrix_host(n_out: int, n_in: int, norm: str | None = "ortho") -> np.ndarray:
    out = np.empty((n_out, n_in), np.float32)
    check(lib().ap_dct_matrix_host(int(n_out), int(n_in), 1 if norm == "ortho" else 0,
                                   out.ctypes.data))
    return out


def _mel_scale_host(fn_name: str, arr, htk: bool) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(arr, dtype=np.float64))
    out = np.empty_like(a)
    check(getattr(lib(), fn_name)(a.ctypes.data, a.size, int(bool(htk)), out.ctypes.data))
    return out


# --------------------------------------------------------------------------
# `_ext`: the reference extension's Python-visible surface (bindings.cpp:15-368)
# --------------------------------------------------------------------------
class _Ext:
    """Same entry points, argument order and defaults as the reference's nanobind
    module; tensors are torch tensors in HBM; `stream` = a torch.cuda.Stream or None."""

    @staticmethod
    def _stream(stream, device):
        return stream.cuda_stream if stream is not None else stream_ptr(device)

    def overlap_add(self, frames, window, hop_length, output_length, stream=None):
        import torch

        frames = to_device_f32(frames)
        if frames.ndim != 3:
            raise ValueError("frames must be 3D (batch, n_frames, n_fft)")
        window = to_device_f32(window, frames.device)
        if window.ndim != 1:
            raise ValueError("window must be 1D")
        B, T, N = frames.shape
        if window.shape[0] != N:
            raise ValueError("Window length must match frame length (n_fft)")
        if hop_length <= 0:
            raise ValueError("hop_length must be positive")
        if output_length <= 0:
            raise ValueError("output_length must be positive")
        out = torch.empty((B, output_length), dtype=torch.float32, device=frames.device)
        check(dlib(frames.device).ap_overlap_add_f32(ptr(frames), ptr(window), B, T, N, int(hop_length), 0,
                                       int(output_length), ptr(out),
                                       self._stream(stream, frames.device)))
        return out

    def frame_signal(self, signal, frame_length, hop_length, stream=None):
        import torch

        signal = to_device_f32(signal)
        one_d = signal.ndim == 1
        if one_d:
            signal = signal[None, :]
        if signal.ndim != 2:
            raise ValueError("signal must be 1D or 2D")
        B, L = signal.shape
        if frame_length <= 0:
            raise ValueError("frame_length must be positive")
        if hop_length <= 0:
            raise ValueError("hop_length must be positive")
        if L < frame_length:
            raise ValueError(
                f"Signal length ({L}) must be >= frame_length ({frame_length}). "
                f"Consider padding the signal."
            )
        T = 1 + (L - frame_length) // hop_length
        out = torch.empty((B, T, frame_length), dtype=torch.float32, device=signal.device)
        check(dlib(signal.device).ap_frame_f32(ptr(signal), B, L, int(frame_length), int(hop_length), ptr(out),
                                 self._stream(stream, signal.device)))
        return out[0] if one_d else out

    def pad_signal(self, signal, pad_length, mode="constant", stream=None):
        import torch

        signal = to_device_f32(signal)
        if signal.ndim != 2:
            raise ValueError("signal must be 2D (batch, samples)")
        if mode not in PAD_MODES:
            raise ValueError(f"Unknown pad mode: '{mode}'. Supported: constant, edge, reflect")
        if pad_length < 0:
            raise ValueError("pad_length must be non-negative")
        if pad_length == 0:
            return signal                                      # pad_signal.cpp:149-151
        B, L = signal.shape
        out = torch.empty((B, L + 2 * pad_length), dtype=torch.float32, device=signal.device)
        check(dlib(signal.device).ap_pad_f32(ptr(signal), B, L, int(pad_length), PAD_MODES[mode], ptr(out),
                               self._stream(stream, signal.device)))
        return out

    def generate_window(self, window_type, length, periodic=True, stream=None):
        import torch

        w = generate_window_host(window_type, length, periodic)
        t = torch.from_numpy(w)
        return t.to(require_device()) if torch.cuda.is_available() else t

    def hz_to_mel(self, frequencies, htk=False, stream=None):
        return _mel_scale_host("ap_hz_to_mel_host", frequencies, htk).astype(np.float32)

    def mel_to_hz(self, mels, htk=False, stream=None):
        return _mel_scale_host("ap_mel_to_hz_host", mels, htk).astype(np.float32)

    def mel_filterbank(self, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False,
                       norm="slaney", stream=None):
        import torch

        t = torch.from_numpy(mel_filterbank_host(sr, n_fft, n_mels, fmin, fmax, htk, norm))
        return t.to(require_device()) if torch.cuda.is_available() else t

    def dct(self, x, n=-1, axis=-1, norm="ortho", stream=None):
        from .mfcc import dct as _dct

        return _dct(x, n=None if n is None or n < 0 else n, axis=axis, norm=norm or None)

    def get_dct_matrix(self, n_out, n_in, norm="ortho", stream=None):
        import torch

        t = torch.from_numpy(dct_matrix_host(n_out, n_in, norm))
        return t.to(require_device()) if torch.cuda.is_available() else t

    def autocorrelation(self, signal, max_lag=-1, normalize=True, center=True, stream=None):
        """bindings.cpp:224-231 (max_lag < 0: all lags), the call pitch.py:59-64 makes."""
        from .pitch import autocorrelation as _acf

        return _acf(signal, max_lag=None if max_lag is None or max_lag < 0 else max_lag, normalize=normalize,
                    center=center)

    # spectral.cpp:8-257 — one statistics kernel (ap_spectral_stats_f32) behind all four.  The native reference
    # guards its quotients with max(sum, 1e-10) (spectral.cpp:47,114) and divides the flatness by the bare mean
    # (:250) where its Python twins add 1e-10 (features.py:128,262,437): the same to float32 rounding unless a
    # whole column is below 1e-9, where this mirror follows the Python twins.
    def spectral_centroid(self, S, frequencies, stream=None):
        from .features import spectral_centroid as _f

        return _f(S=S, freq=frequencies)

    def spectral_bandwidth(self, S, frequencies, centroid, p=2.0, stream=None):
        """bindings.cpp:398-405: an empty `centroid` means "compute it" (spectral.cpp:88-96)."""
        from .features import spectral_bandwidth as _f

        if centroid is not None and int(np.prod(tuple(centroid.shape))) == 0:
            centroid = None
        return _f(S=S, freq=frequencies, centroid=centroid, p=p)

    def spectral_rolloff(self, S, frequencies, roll_percent=0.85, stream=None):
        """bindings.cpp:430-436, the call features.py:352 makes."""
        from .features import spectral_rolloff as _f

        return _f(S=S, freq=frequencies, roll_percent=roll_percent)

    def spectral_flatness(self, S, amin=1e-10, stream=None):
        from .features import spectral_flatness as _f

        return _f(S=S, amin=amin)

    def resample_fft(self, signal, num_samples, stream=None):
        """resample.cpp:9-98: full complex spectrum cut or zero-filled in the middle, real part of the inverse,
        times num_samples / n.  That is scipy.signal.resample (our `resample(res_type="fft")` engine) except
        when shrinking to an even length M: SciPy folds the bins +M/2 and -M/2 together, the native code keeps
        only -M/2 — half the Nyquist term, Re X[M/2] (-1)^j / n, taken off here.  Unreachable from the reference's
        public API (resample.py never calls `_ext`) and only shape-tested there (test_cpp_extension.py:91-124)."""
        import torch

        from .resample import _resample_fft_length

        x = to_device_f32(signal)
        if x.ndim not in (1, 2):
            raise ValueError("signal must be 1-dimensional (samples,) or 2-dimensional (batch, samples)")
        if num_samples <= 0:
            raise ValueError("num_samples must be positive")
        n = x.shape[-1]
        if num_samples == n:
            return x
        out = _resample_fft_length(x, int(num_samples))
        if num_samples < n and num_samples % 2 == 0:
            j = np.arange(n, dtype=np.float64)
            c = torch.from_numpy(np.cos(2.0 * np.pi * ((num_samples // 2) * j % n) / n).astype(np.float32)).to(x.device)
            nyq = (x * c).sum(-1, keepdim=True) / n
            sign = torch.ones(num_samples, dtype=torch.float32, device=x.device)
            sign[1::2] = -1.0
            out = out - nyq * sign
        return out

    def resample(self, signal, orig_sr, target_sr, fix=True, scale=False, stream=None):
        """resample.cpp:100-149: length round(n ratio) (fix) or ceil(n ratio), then resample_fft, times ratio if
        `scale`."""
        if orig_sr <= 0 or target_sr <= 0:
            raise ValueError("Sample rates must be positive")
        x = to_device_f32(signal)
        if orig_sr == target_sr:
            return x
        ratio = float(target_sr) / float(orig_sr)
        n = x.shape[-1]
        m = int(np.floor(n * ratio + 0.5)) if fix else int(np.ceil(n * ratio))      # std::round: half away from zero
        out = self.resample_fft(x, m)
        return out * np.float32(ratio) if scale else out


_ext: _Ext | None = _Ext() if HAS_HIP_EXT else None

__all__ = ["_ext", "HAS_HIP_EXT", "lib", "check"]
