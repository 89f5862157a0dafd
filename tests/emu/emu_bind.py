"""ctypes bindings for tests/emu/libapemu.so (TEST INFRASTRUCTURE ONLY).

The emulator runs the product's kernel source on the CPU (see emu_shim.h).  It is
built on demand with g++ and is never imported by the product package.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "mlx-audio-primitives_amd", "csrc")
LIB = os.path.join(HERE, "libapemu.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.c_int64
_int = ctypes.c_int


def build(force=False, sanitize=False):
    srcs = [os.path.join(HERE, "emu_lib.cpp"), os.path.join(CSRC, "host_builders.cpp")]
    deps = srcs + [os.path.join(HERE, "emu_shim.h")] + [
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")
    ]
    if not force and os.path.exists(LIB):
        if os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps):
            return LIB
    cmd = ["g++", "-O2", "-std=c++20", "-fPIC", "-shared", "-pthread", "-o", LIB] + srcs
    if sanitize:
        cmd[1:1] = ["-fsanitize=address,undefined", "-g"]
    subprocess.check_call(cmd)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.emu_last_error.restype = ctypes.c_char_p
        _lib.ap_twiddle_table_host.argtypes = [_int, _f32p]
    return _lib


def _p(a):
    return a.ctypes.data_as(_f32p)


def _check(rc):
    if rc != 0:
        raise ValueError(lib().emu_last_error().decode())


def twiddles(n_fft):
    tw = np.empty(2 * n_fft, np.float32)
    _check(lib().ap_twiddle_table_host(n_fft, _p(tw)))
    return tw


def n_frames(L, n_fft, hop, center):
    Lp = L + (2 * (n_fft // 2) if center else 0)
    return 1 + (Lp - n_fft) // hop


def stft(y, n_fft, hop, window, center=True, pad_mode=0):
    y = np.ascontiguousarray(y, np.float32)
    B, L = y.shape
    T = n_frames(L, n_fft, hop, center)
    out = np.zeros((B, n_fft // 2 + 1, T, 2), np.float32)
    window = np.ascontiguousarray(window, np.float32)
    tw = twiddles(n_fft)
    _check(lib().emu_stft_f32(_p(y), _i64(B), _i64(L), n_fft, hop, _p(window), _p(tw),
                              int(center), pad_mode, _i64(T), _p(out)))
    return out[..., 0] + 1j * out[..., 1]


def stft16(y, hop, window, center=True, pad_mode=0, Ts=None, grid_cap=2, misalign=0, force_unaligned=False):
    """kernels_stft16.h: (B, 1025, T) complex with rows Ts apart, written at an offset of `misalign`
    complex values from a 128-byte aligned base.  Returns (spectrum, whole buffer, aligned flag)."""
    y = np.ascontiguousarray(y, np.float32)
    B, L = y.shape
    T = n_frames(L, 2048, hop, center)
    Ts = T if Ts is None else Ts
    n = B * 1025 * Ts
    raw = np.full(2 * (n + 64) + 32, np.float32(-777.0), np.float32)
    off = (-raw.ctypes.data // 4) % 32 + 2 * misalign          # floats to a 128-byte boundary
    out = raw[off:off + 2 * n]
    window = np.ascontiguousarray(window, np.float32)
    tw = twiddles(2048)
    rc = lib().emu_stft16_f32(_p(y), _i64(B), _i64(L), hop, _p(window), _p(tw), int(center), pad_mode,
                              _i64(T), _i64(Ts), _p(out), grid_cap, int(force_unaligned))
    if rc < 0:
        _check(rc)
    full = out.reshape(B, 1025, Ts, 2)
    S = full[:, :, :T, 0] + 1j * full[:, :, :T, 1]
    return S, raw, off, rc


def mel_plan(fb):
    fb = np.ascontiguousarray(fb, np.float32)
    M, F = fb.shape
    L = lib()
    L.ap_mel_plan_words.restype = _i64
    words = L.ap_mel_plan_words(_p(fb), M, F)
    plan = np.zeros(words, np.int32)
    desc = np.zeros(16, np.int32)
    _check(L.ap_mel_plan_host(_p(fb), M, F, plan.ctypes.data_as(_i32p), desc.ctypes.data_as(_i32p)))
    return plan, desc


def melspec(y, n_fft, hop, window, fb, center=True, pad_mode=0, power=2.0, banded=True,
            force_generic=False, return_max=False, tile_kernel=False):
    y = np.ascontiguousarray(y, np.float32)
    B, L = y.shape
    T = n_frames(L, n_fft, hop, center)
    fb = np.ascontiguousarray(fb, np.float32)
    M = fb.shape[0]
    out = np.zeros((B, M, T), np.float32)
    window = np.ascontiguousarray(window, np.float32)
    tw = twiddles(n_fft)
    if banded or force_generic:
        plan, desc = mel_plan(fb)
        if not banded:
            desc[0] = 0                      # dense contraction (no band spans, no parts)
        if force_generic:
            desc[0] |= 256                   # AP_PLAN_FORCE_GENERIC: keep the generic LDS engine
        if tile_kernel:
            desc[0] |= 512                   # emulator-only: n_fft=2048 tile kernel instead of the run kernel
        plan_p, desc_p = plan.ctypes.data_as(_i32p), desc.ctypes.data_as(_i32p)
    else:
        plan_p = desc_p = None
    key = ctypes.c_uint32(0)
    _check(lib().emu_melspec_f32(_p(y), _i64(B), _i64(L), n_fft, hop, _p(window), _p(tw),
                                 int(center), pad_mode, _i64(T), _p(fb), plan_p, desc_p, M,
                                 ctypes.c_float(power), _p(out), ctypes.byref(key) if return_max else None))
    if return_max:
        k = key.value                               # order-preserving key -> float (ap_fkey_inv)
        u = (k & 0x7FFFFFFF) if (k & 0x80000000) else (~k & 0xFFFFFFFF)
        return out, np.array([u], np.uint32).view(np.float32)[0]
    return out


def mel_wave_geometry(fb, B=4, L=22050, hop=512):
    """(partial_stride, lds_bytes, n_slots, grid) of the n_fft=2048 mel wave kernel, or None."""
    fb = np.ascontiguousarray(fb, np.float32)
    plan, desc = mel_plan(fb)
    out = np.zeros(4, np.int32)
    rc = lib().emu_mel_wave_geometry(_p(fb), plan.ctypes.data_as(_i32p), desc.ctypes.data_as(_i32p),
                                     fb.shape[0], _i64(B), _i64(L), hop, out.ctypes.data_as(_i32p))
    return None if rc == 1 else tuple(int(v) for v in out)


def mel_wave_partial_stride(fb):
    return mel_wave_geometry(fb)[0]


def irfft_frames(S, n_fft):
    S = np.asarray(S)
    B, F, T = S.shape
    Si = np.ascontiguousarray(np.stack([S.real, S.imag], -1), np.float32)
    frames = np.zeros((B, T, n_fft), np.float32)
    tw = twiddles(n_fft)
    _check(lib().emu_irfft_frames_f32(_p(Si), _i64(B), _i64(T), n_fft, _p(tw), _p(frames)))
    return frames


def overlap_add(frames, window, hop, out_len, out_offset=0):
    frames = np.ascontiguousarray(frames, np.float32)
    B, T, N = frames.shape
    window = np.ascontiguousarray(window, np.float32)
    out = np.zeros((B, out_len), np.float32)
    _check(lib().emu_overlap_add_f32(_p(frames), _p(window), _i64(B), _i64(T), N, hop,
                                     _i64(out_offset), _i64(out_len), _p(out)))
    return out


def pad(x, pad_len, mode):
    x = np.ascontiguousarray(x, np.float32)
    B, L = x.shape
    out = np.zeros((B, L + 2 * pad_len), np.float32)
    _check(lib().emu_pad_f32(_p(x), _i64(B), _i64(L), _i64(pad_len), mode, _p(out)))
    return out


def frame(x, frame_length, hop):
    x = np.ascontiguousarray(x, np.float32)
    B, L = x.shape
    T = max(1 + (L - frame_length) // hop, 0)
    out = np.zeros((B, T, frame_length), np.float32)
    _check(lib().emu_frame_f32(_p(x), _i64(B), _i64(L), frame_length, hop, _p(out)))
    return out


def resample_poly_taps(up, down):
    L = lib()
    n = L.ap_resample_poly_ntaps(up, down)
    taps = np.zeros(n, np.float32)
    npr = ctypes.c_int(0)
    _check(L.ap_resample_poly_taps_host(up, down, _p(taps), ctypes.byref(npr)))
    return taps, npr.value


def resample_poly(x, up, down, two_outputs=True):
    """two_outputs: let the two-outputs-per-thread decimator (ap_resample_decim2_kernel) serve the shapes it applies to."""
    lib().emu_set_decim2(int(two_outputs))
    x = np.ascontiguousarray(x, np.float32)
    B, L = x.shape
    taps, npr = resample_poly_taps(up, down)
    n_out = (L * up + down - 1) // down
    out = np.zeros((B, n_out), np.float32)
    _check(lib().emu_resample_poly_f32(_p(x), _i64(B), _i64(L), up, down, _p(taps), len(taps), npr,
                                       _i64(n_out), _p(out)))
    return out


def resample_linear(x, n_out, scale=1.0):
    x = np.ascontiguousarray(x, np.float32)
    B, L = x.shape
    out = np.zeros((B, n_out), np.float32)
    _check(lib().emu_resample_linear_f32(_p(x), _i64(B), _i64(L), _i64(n_out),
                                         ctypes.c_double(scale), _p(out)))
    return out


def gl_project(mode, S, angles=None, R=None, momentum=0.99, tprev=None):
    S = np.ascontiguousarray(S, np.float32)
    B, F, T = S.shape
    reb = np.zeros((B, F, T, 2), np.float32)
    tp = np.zeros((B, F, T, 2), np.float32) if tprev is None else \
        np.ascontiguousarray(np.stack([tprev.real, tprev.imag], -1), np.float32)
    ang_p = _p(np.ascontiguousarray(angles, np.float32)) if angles is not None else None
    if R is not None:
        Rr = np.ascontiguousarray(np.stack([R.real, R.imag], -1), np.float32)
        R_p, TR = _p(Rr), R.shape[-1]
    else:
        R_p, TR = None, 0
    _check(lib().emu_gl_project_f32(mode, _p(S), ang_p, R_p, _i64(TR), _i64(B * F), _i64(T),
                                    ctypes.c_float(momentum), _p(tp), _p(reb)))
    return reb[..., 0] + 1j * reb[..., 1], tp[..., 0] + 1j * tp[..., 1]


def to_db(S, coef=10.0, amin=1e-10, ref=1.0, ref_is_max=False, top_db=80.0):
    S = np.ascontiguousarray(S, np.float32)
    out = np.zeros_like(S)
    _check(lib().emu_to_db_f32(_p(S), _i64(S.size), ctypes.c_float(coef), ctypes.c_float(amin),
                               ctypes.c_float(ref), int(ref_is_max),
                               ctypes.c_float(-1.0 if top_db is None else top_db), _p(out)))
    return out


def dct(x, C, outer, n_in, inner, row_scale=None, db=None, wide=False):
    """db = (coef, amin, ref_value, top_db or None): fused power_to_db + DCT (ap_db_dct_f32);
    wide: the 64-bit-offset instantiation (tensors of 2^31 elements and more on the device)."""
    lib().emu_set_dct_wide(int(wide))
    x = np.ascontiguousarray(x, np.float32)
    C = np.ascontiguousarray(C, np.float32)
    n_out = C.shape[0]
    out = np.zeros(outer * n_out * inner, np.float32)
    rs = _p(np.ascontiguousarray(row_scale, np.float32)) if row_scale is not None else None
    f = ctypes.c_float
    coef, amin, ref, top = db if db is not None else (0.0, 0.0, 0.0, None)
    _check(lib().emu_dct_f32(_p(x), _p(C), rs, _i64(outer), n_in, _i64(inner), n_out,
                             int(db is not None), f(coef), f(amin), f(ref),
                             f(-1.0 if top is None else top), _p(out)))
    return out


def istft_fused(S, hop, window, out_len, out_offset=None, grid_cap=0):
    """Fused irfft + overlap-add of an n_fft = 2048 (B, 1025, T) or n_fft = 1024 (B, 513, T) spectrum
    (ap_irfft2048_wave_kernel<1> / ap_istft1024_wave_kernel)."""
    S = np.ascontiguousarray(S, np.complex64)
    B, F, T = S.shape
    n_fft = 2 * (F - 1)
    if out_offset is None:
        out_offset = n_fft // 2
    Sv = np.ascontiguousarray(S.view(np.float32))
    out = np.zeros((B, out_len), np.float32)
    window = np.ascontiguousarray(window, np.float32)
    tw = twiddles(n_fft)
    if n_fft in (512, 400, 256):
        _check(lib().emu_istft8_fused_f32(_p(Sv), _i64(B), _i64(T), n_fft, hop, _p(window), _p(tw), _i64(out_offset),
                                          _i64(out_len), grid_cap, _p(out)))
        return out
    fn = lib().emu_istft_fused_f32 if n_fft == 2048 else lib().emu_istft1024_fused_f32
    _check(fn(_p(Sv), _i64(B), _i64(T), hop, _p(window), _p(tw), _i64(out_offset), _i64(out_len), grid_cap,
              _p(out)))
    return out


def _aligned128(a):
    """A copy of `a` whose data starts on a 128-byte boundary."""
    raw = np.empty(a.nbytes + 128, np.uint8)
    off = (-raw.ctypes.data) % 128
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def stft16_gl(y, hop, window, prev, mag, momentum, center=True, pad_mode=0, Ts=None, grid_cap=2, variant=2):
    """kernels_stft16.h with the Griffin-Lim projection in its store phase: returns (raw spectrum, rebuilt) as
    (B, 1025, T) complex arrays; `prev` (B, 1025, T) complex, `mag` (B, 1025, T) float; rows held Ts apart."""
    y = np.ascontiguousarray(y, np.float32)
    B, L = y.shape
    T = n_frames(L, 2048, hop, center)
    Ts = -(-T // 16) * 16 if Ts is None else Ts

    def padded(a, fill):
        buf = np.full((B, 1025, Ts), fill, np.complex64)
        if a is not None:
            buf[:, :, :T] = a
        return _aligned128(buf)

    pv = padded(prev, np.complex64(3e3 - 2e3j))
    out = padded(None, np.complex64(-777 - 777j))
    reb = padded(None, np.complex64(-555 - 555j))
    mag = np.ascontiguousarray(mag, np.float32)
    window = np.ascontiguousarray(window, np.float32)
    tw = twiddles(2048)
    _check(lib().emu_stft16_gl_f32(_p(y), _i64(B), _i64(L), hop, _p(window), _p(tw), int(center), pad_mode, _i64(T), _i64(Ts),
                                   _p(pv), _p(mag), ctypes.c_float(momentum), _p(out), _p(reb), grid_cap, variant))
    return out[:, :, :T].copy(), reb[:, :, :T].copy(), out, reb


def istft16(S, hop, window, out_len, out_offset=1024, grid_cap=0, Ts=None, variant=0):
    """kernels_istft16.h: fused ISTFT of an n_fft = 2048 (B, 1025, T) spectrum held with rows Ts apart."""
    S = np.ascontiguousarray(S, np.complex64)
    B, F, T = S.shape
    assert F == 1025
    Ts = T if Ts is None else Ts
    buf = np.full((B, F, Ts), np.complex64(7e3 + 9e3j), np.complex64)     # padding the kernel must not use
    buf[:, :, :T] = S
    Sv = np.ascontiguousarray(buf.view(np.float32))
    out = np.full((B, out_len), np.float32(-555.0), np.float32)
    window = np.ascontiguousarray(window, np.float32)
    tw = twiddles(2048)
    _check(lib().emu_istft16_f32(_p(Sv), _i64(B), _i64(T), _i64(Ts), hop, _p(window), _p(tw), _i64(out_offset),
                                 _i64(out_len), grid_cap, variant, _p(out)))
    return out


def cfft_split(N):
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().emu_cfft_split(_i64(N), ctypes.byref(a), ctypes.byref(b))
    return (a.value, b.value) if rc == 0 else None


def chirp_tables(N):
    """float64-built Bluestein tables (same construction as mlx-audio-primitives_amd/resample.py)."""
    n = np.arange(N, dtype=np.int64)
    c = np.exp(-1j * np.pi * ((n * n) % (2 * N)) / N)
    M = 1 << int(2 * N - 2).bit_length() if N > 1 else 1
    b = np.zeros(M, np.complex128)
    b[:N] = np.conj(c)
    if N > 1:
        b[M - N + 1:] = np.conj(c[1:][::-1])
    return np.ascontiguousarray(c.astype(np.complex64)), np.ascontiguousarray(np.fft.fft(b).astype(np.complex64)), M


def resample_fft(x, num, force_chirp=False):
    x = np.ascontiguousarray(x, np.float32)
    B, Nx = x.shape
    args, alive, nmax = [], [], max(Nx, num)
    for N in (Nx, num):
        split = None if force_chirp else cfft_split(N)
        if split is not None:
            t1, t2 = twiddles(split[0]), twiddles(split[1])
            alive += [t1, t2]
            args += [_i64(0), _p(t1), _p(t2), None, None]
        else:
            c, spec, M = chirp_tables(N)
            m1, m2 = cfft_split(M)
            t1, t2 = twiddles(m1), twiddles(m2)
            alive += [c, spec, t1, t2]
            args += [_i64(M), _p(t1), _p(t2), _p(c), _p(spec)]
            nmax = max(nmax, M)
    ws = np.zeros(4 * B * nmax, np.float32)
    out = np.zeros((B, num), np.float32)
    _check(lib().emu_resample_fft_chirp_f32(_p(x), _i64(B), _i64(Nx), _i64(num), *args, _p(ws), _p(out)))
    return out


def pcg64_uniform(seed, low, high, n):
    st = np.random.default_rng(seed).bit_generator.state["state"]
    m = (1 << 64) - 1
    out = np.zeros(n, np.float32)
    u64 = ctypes.c_uint64
    _check(lib().emu_pcg64_uniform_f32(u64(st["state"] >> 64), u64(st["state"] & m), u64(st["inc"] >> 64),
                                       u64(st["inc"] & m), ctypes.c_double(low), ctypes.c_double(high),
                                       _i64(n), _p(out)))
    return out


def spectral_from_audio(y, sr, hop, window, center=True, power=1.0, p=2.0, norm=True, roll_percent=0.85,
                        amin=1e-10, want=("centroid", "bandwidth", "rolloff")):
    """ap_spec2048_run_kernel (n_fft = 2048) on the CPU: {name: (B, T)}."""
    y = np.ascontiguousarray(y, np.float32)
    B, L = y.shape
    T = n_frames(L, 2048, hop, center)
    window = np.ascontiguousarray(window, np.float32)
    tw = twiddles(2048)
    freq = np.linspace(0, sr / 2.0, 1025).astype(np.float32)
    outs = {k: np.zeros((B, T), np.float32) for k in want}
    g = lambda k: _p(outs[k]) if k in outs else None  # noqa: E731
    f = ctypes.c_float
    _check(lib().emu_spectral_audio_f32(_p(y), _i64(B), _i64(L), hop, _p(window), _p(tw), int(center), _i64(T),
                                        _p(freq), f(power), f(p), int(norm), f(roll_percent), f(amin),
                                        g("centroid"), g("bandwidth"), g("rolloff"), g("flatness")))
    return outs


def lds_overruns():
    """Launches since the library was loaded that wrote past the dynamic LDS their host code asked for."""
    return int(lib().emu_lds_overrun_count())
