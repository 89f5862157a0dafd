"""CPU: run the product's *kernel source* (kernels_generic.h, fft_lds.h) through the
SIMT emulator in tests/emu and compare with the oracle.  Catches indexing, twiddle,
radix-plan and pad-remap bugs before any GPU time is spent.  The emulator is test
infrastructure; the product never loads it."""

import os
import sys

import numpy as np
import pytest

from oracle import audio_oracle as ao

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu_bind as eb  # noqa: E402

PM = {"constant": 0, "edge": 1, "reflect": 2}

STFT_CASES = [
    # n_fft, hop, L, B, pad_mode, center   (radix plans: 16*16, 16*16*4, 8*5*5, 8*4, 3*5, 3*3*3, 11, 7*11, 16*16*16)
    (512, 128, 4000, 2, "constant", True),
    (2048, 512, 6000, 1, "reflect", True),
    # 3 clips x 22 frames = 9 groups on 2 workgroups: carried (sector-aligned) row windows, a clip
    # change inside a workgroup's stretch and a stretch that ends mid-clip
    (2048, 512, 10752, 3, "constant", True),
    (2048, 512, 9300, 2, "constant", False),
    # n_fft = 1024 wave kernel: 3 clips x 22 frames (carries, clip change in a stretch), no centring
    (1024, 256, 5376, 3, "constant", True),
    (1024, 300, 6100, 2, "constant", False),
    (1024, 256, 4000, 2, "reflect", True),       # index-remapped edge frames in the wave kernel
    (1024, 255, 3000, 1, "constant", True),      # centred frames at an odd hop
    (400, 160, 3000, 3, "constant", True),
    (64, 16, 500, 1, "edge", True),
    (30, 7, 400, 2, "constant", False),
    (27, 5, 300, 1, "constant", True),
    (22, 11, 300, 1, "reflect", True),
    (154, 30, 800, 1, "constant", True),
    (8192, 2048, 20000, 1, "constant", True),
    (2, 1, 40, 1, "constant", True),
    (1, 1, 10, 1, "constant", True),
]


@pytest.mark.parametrize("n_fft,hop,L,B,pad_mode,center", STFT_CASES)
def test_emu_stft(n_fft, hop, L, B, pad_mode, center):
    rng = np.random.default_rng(n_fft + hop)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", n_fft, n_fft)
    S = eb.stft(y, n_fft, hop, win, center, PM[pad_mode])
    R = ao.stft(y, n_fft=n_fft, hop_length=hop, center=center, pad_mode=pad_mode)
    assert S.shape == R.shape
    np.testing.assert_allclose(S, R, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("hop,L,B,pad_mode,center,Ts,grid_cap,misalign,force_unaligned", [
    # contiguous rows (Ts = T): line-aligned windows with register carries; 3 clips x 22 frames = 6 groups
    # on 2 workgroups (a clip change inside a stretch, a stretch that ends mid-clip), T odd and even
    (512, 10752, 3, "constant", True, None, 2, 0, False),
    (512, 20000, 2, "constant", True, None, 1, 3, False),      # 40 frames: 3 groups per clip, one workgroup
    (512, 20000, 2, "constant", True, None, 5, 0, False),      # more workgroups than clips: stretches start mid-clip
    (512, 9300, 2, "constant", False, None, 2, 5, False),
    (512, 6000, 1, "reflect", True, None, 2, 0, False),
    (300, 9000, 2, "constant", True, None, 2, 1, False),
    (255, 5000, 1, "constant", True, None, 2, 0, False),       # centred frames at an odd hop: index-remapped loads
    # padded rows (Ts a multiple of 16, aligned base): whole lines, no carries
    (512, 10752, 3, "constant", True, 32, 2, 0, False),
    (512, 20000, 2, "edge", True, 48, 3, 0, False),
    # ... and the carry path on the same layout (phi = 0 everywhere: every store lags a whole group)
    (512, 20000, 2, "constant", True, 48, 3, 0, True),
    (512, 20000, 1, "constant", True, 45, 2, 7, False),        # padded, but not to a line
    # one group per workgroup, many groups per workgroup, odd hop (index-remapped loads) on padded rows
    (512, 20000, 2, "constant", True, 48, 6, 0, False),
    (512, 60000, 1, "constant", False, 128, 1, 0, False),
    (255, 9000, 2, "constant", True, 48, 2, 0, False),
    # padded rows on the whole-group tile (3; AP_STFT16_T2=1)
    (512, 20000, 2, "edge", True, 48, 3, 0, 3),
    (512, 60000, 1, "constant", False, 128, 1, 0, 3),
    (512, 10752, 3, "constant", True, 32, 2, 0, 3),
])
def test_emu_stft16(hop, L, B, pad_mode, center, Ts, grid_cap, misalign, force_unaligned):
    """kernels_stft16.h (n_fft = 2048, 16 frames per group, 128-byte row windows) on the CPU."""
    rng = np.random.default_rng(hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", 2048, 2048)
    S, raw, off, aligned = eb.stft16(y, hop, win, center, PM[pad_mode], Ts, grid_cap, misalign, force_unaligned)
    R = ao.stft(y, n_fft=2048, hop_length=hop, center=center, pad_mode=pad_mode)
    assert S.shape == R.shape
    np.testing.assert_allclose(S, R, rtol=1e-4, atol=1e-4)
    T = R.shape[-1]
    Tr = T if Ts is None else Ts
    assert aligned == int(Tr % 16 == 0 and misalign == 0 and force_unaligned != 1)
    # nothing outside the T frames of every row was touched (row padding, guard floats either side)
    n = B * 1025 * Tr
    assert np.all(raw[:off] == -777.0) and np.all(raw[off + 2 * n:] == -777.0)
    assert np.all(raw[off:off + 2 * n].reshape(B, 1025, Tr, 2)[:, :, T:] == -777.0)


@pytest.mark.parametrize("hop,L,B,momentum,grid_cap", [
    (512, 10752, 3, 0.99, 2),       # T = 22: a full group and one with 6 frames per clip (odd pairs at the clip's end: T even)
    (512, 11300, 2, 0.99, 1),       # T = 23: the last pair of every row has one frame
    (512, 20000, 2, 0.0, 3),        # no momentum
    (256, 9000, 1, 0.5, 2),
])
def test_emu_stft16_griffinlim_projection(hop, L, B, momentum, grid_cap):
    """kernels_stft16.h, GL = 1 (8-byte accesses) and GL = 2 (two frames per thread, 16-byte accesses): raw spectrum and
    rebuilt = S unit(raw) + m (S unit(raw) - S unit(prev)) (griffinlim.py:156-178), nothing written outside the T frames."""
    rng = np.random.default_rng(hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", 2048, 2048)
    R = ao.stft(y, n_fft=2048, hop_length=hop)
    T = R.shape[-1]
    prev = (rng.standard_normal(R.shape) + 1j * rng.standard_normal(R.shape)).astype(np.complex64)
    mag = rng.random(R.shape).astype(np.float32)
    outs = {}
    for variant in (1, 2):
        raw, reb, raw_full, reb_full = eb.stft16_gl(y, hop, win, prev, mag, momentum, grid_cap=grid_cap, variant=variant)
        np.testing.assert_allclose(raw, R, rtol=1e-4, atol=1e-4)
        unit = lambda z: z / np.maximum(np.abs(z), 1e-30)
        P = mag * unit(raw.astype(np.complex128))
        want = P + momentum * (P - mag * unit(prev.astype(np.complex128)))
        np.testing.assert_allclose(reb, want, rtol=2e-3, atol=2e-3)          # v_rsq_f32-grade reciprocal square roots
        assert np.all(raw_full[:, :, T:] == np.complex64(-777 - 777j)) and np.all(reb_full[:, :, T:] == np.complex64(-555 - 555j))
        outs[variant] = (raw, reb)
    assert np.array_equal(outs[1][0], outs[2][0]) and np.array_equal(outs[1][1], outs[2][1])      # bit for bit


@pytest.mark.parametrize("sr,n_fft,hop,M,L,B,power", [
    (22050, 2048, 512, 128, 9000, 2, 2.0),
    (16000, 400, 160, 80, 5000, 3, 2.0),
    (22050, 1024, 256, 40, 5000, 1, 1.0),
    (22050, 512, 128, 64, 4000, 1, 1.5),
])
def test_emu_melspec(sr, n_fft, hop, M, L, B, power):
    rng = np.random.default_rng(M)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", n_fft, n_fft)
    fb = ao.mel_filterbank(sr, n_fft, M)
    R = ao.melspectrogram(y, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=M, power=power)
    banded = eb.melspec(y, n_fft, hop, win, fb, power=power, banded=True, force_generic=True)
    dense = eb.melspec(y, n_fft, hop, win, fb, power=power, banded=False, force_generic=True)
    np.testing.assert_allclose(banded, R, rtol=1e-4, atol=1e-4)
    # skipping the zeros outside each filter's span must not change a single bit (same engine)
    np.testing.assert_array_equal(banded, dense)
    # no plan at all: dense contraction on whichever engine serves this n_fft
    np.testing.assert_allclose(eb.melspec(y, n_fft, hop, win, fb, power=power, banded=False), R,
                               rtol=1e-4, atol=1e-4)
    # with the plan: n_fft=2048 -> wave kernel, 400/512/1024 -> compile-time engine with the
    # LDS parts contraction
    planned = eb.melspec(y, n_fft, hop, win, fb, power=power, banded=True)
    np.testing.assert_allclose(planned, R, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("sr,M,L,B,power,pad_mode,kw", [
    (22050, 128, 9000, 2, 2.0, "constant", {}),
    (22050, 80, 30000, 1, 1.0, "reflect", dict(fmin=300.0, fmax=8000.0)),
    (16000, 40, 2100, 3, 2.0, "edge", dict(htk=True)),
    (22050, 128, 17 * 512, 1, 1.5, "constant", dict(norm=None)),
])
def test_emu_wave_kernel(sr, M, L, B, power, pad_mode, kw):
    """kernels_wave.h (n_fft=2048, wave per frame, DPP quad radix-4, plan contraction) on the CPU."""
    rng = np.random.default_rng(L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", 2048, 2048)
    fb = ao.mel_filterbank(sr, 2048, M, **kw)
    plan, desc = eb.mel_plan(fb)
    assert desc[0] & 2 and desc[1] == M
    A, amax = eb.melspec(y, 2048, 512, win, fb, power=power, pad_mode=PM[pad_mode], return_max=True)
    R = ao.melspectrogram(y, sr=sr, n_fft=2048, hop_length=512, n_mels=M, power=power,
                          pad_mode=pad_mode, **kw)
    np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)
    assert amax == A.max()            # the key the kernel raises for mfcc's top_db clip
    # constant padding with power 2 / 1 runs on the run kernel (kernels_mel2048.h); the
    # tile kernel (kernels_wave.h) keeps serving the other shapes and must agree
    A2, amax2 = eb.melspec(y, 2048, 512, win, fb, power=power, pad_mode=PM[pad_mode], return_max=True,
                           tile_kernel=True)
    np.testing.assert_allclose(A2, R, rtol=1e-4, atol=1e-4)
    assert amax2 == A2.max()


@pytest.mark.parametrize("sr,M,kw", [
    (22050, 20, dict(fmax=2000.0)),      # 34 partial-sum slots: fewer than the 64 lanes of the max reduction
    (48000, 8, dict(fmax=1000.0)),       # 8 slots
    (22050, 10, {}),                     # rows of up to 30 parts (max_row_parts > 4)
    (22050, 136, {}),                    # rows beyond the two per lane
    (22050, 160, dict(norm=None)),
])
def test_emu_wave_kernel_unusual_filterbanks(sr, M, kw):
    """Branches of the n_fft=2048 mel kernel the default 128-filter bank never takes."""
    rng = np.random.default_rng(M)
    y = rng.standard_normal((2, 6000)).astype(np.float32)
    win = ao.padded_window("hann", 2048, 2048)
    fb = ao.mel_filterbank(sr, 2048, M, **kw)
    R = ao.melspectrogram(y, sr=sr, n_fft=2048, hop_length=512, n_mels=M, **kw)
    for tile_kernel in (False, True):
        A, amax = eb.melspec(y, 2048, 512, win, fb, return_max=True, tile_kernel=tile_kernel)
        A0 = eb.melspec(y, 2048, 512, win, fb, tile_kernel=tile_kernel)
        np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)
        np.testing.assert_array_equal(A, A0)
        assert amax == A.max()
    # every wave stages its 64 lane maxima in its own partial-sum region
    assert eb.mel_wave_partial_stride(fb) >= 64


@pytest.mark.parametrize("hop", [256, 1024, 500, 128])
def test_emu_wave_kernel_hops(hop):
    """Frame-to-frame register reuse of the overlapping samples (hop 256 / 512 / 1024) against the
    full-frame loads every other hop takes; two clips so a clip change falls inside a stretch."""
    rng = np.random.default_rng(hop)
    y = rng.standard_normal((2, 7000)).astype(np.float32)
    win = ao.padded_window("hann", 2048, 2048)
    fb = ao.mel_filterbank(22050, 2048, 64)
    R = ao.melspectrogram(y, sr=22050, n_fft=2048, hop_length=hop, n_mels=64)
    for tile_kernel in (False, True):
        A = eb.melspec(y, 2048, hop, win, fb, tile_kernel=tile_kernel)
        np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("hop,M,L,B,power", [(256, 128, 9000, 2, 2.0), (128, 80, 5000, 1, 1.0),
                                              (512, 64, 12000, 3, 2.0), (300, 40, 7000, 2, 1.5)])
def test_emu_wave512_kernel(hop, M, L, B, power):
    """kernels_wave512.h (n_fft=1024: 8 x 8 x 8 wave-per-frame transform, output run in registers):
    register reuse across frames at hop 128 / 256 / 512, full loads otherwise, clip changes."""
    rng = np.random.default_rng(hop + M)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", 1024, 1024)
    fb = ao.mel_filterbank(22050, 1024, M)
    A, amax = eb.melspec(y, 1024, hop, win, fb, power=power, return_max=True)
    R = ao.melspectrogram(y, sr=22050, n_fft=1024, hop_length=hop, n_mels=M, power=power)
    np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)
    assert amax == A.max()


@pytest.mark.parametrize("n_fft,hop,L,B", [
    (512, 128, 4000, 2), (2048, 512, 9000, 1), (400, 160, 3000, 2), (27, 5, 300, 1),
    (30, 7, 300, 2), (8192, 2048, 20000, 1),
    # 3 clips x 22 frames on 2 workgroups: carried sector-aligned windows across groups, a clip
    # change inside a stretch, odd and even T
    (2048, 512, 10752, 3), (2048, 512, 11300, 2),
])
def test_emu_istft(n_fft, hop, L, B):
    rng = np.random.default_rng(n_fft)
    y = rng.standard_normal((B, L)).astype(np.float32)
    S = ao.stft(y, n_fft=n_fft, hop_length=hop)
    fr = eb.irfft_frames(S, n_fft)
    ref = np.fft.irfft(np.transpose(S, (0, 2, 1)).astype(np.complex128), n=n_fft, axis=-1)
    np.testing.assert_allclose(fr, ref, atol=2e-6)
    win = ao.padded_window("hann", n_fft, n_fft)
    out = eb.overlap_add(fr, win, hop, L, out_offset=n_fft // 2)
    np.testing.assert_allclose(out, ao.istft(S, hop_length=hop, n_fft=n_fft, length=L), atol=1e-5)
    if n_fft % hop == 0 and n_fft // hop >= 4:
        assert np.max(np.abs(out - y)) < 1e-5        # README.md:118


@pytest.mark.parametrize("hop,L,B,grid_cap", [(512, 10752, 3, 2), (512, 11300, 2, 3), (1024, 30000, 2, 2),
                                              (256, 9000, 1, 2), (512, 6000, 1, 0)])
def test_emu_istft_fused(hop, L, B, grid_cap):
    """Fused irfft + overlap-add kernel: carries across the groups of a stretch, warm-up group of a
    stretch that starts inside a clip, clip change inside a stretch, tail after the last group."""
    rng = np.random.default_rng(hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    S = ao.stft(y, n_fft=2048, hop_length=hop)
    win = ao.padded_window("hann", 2048, 2048)
    for length in (L, L - 700):
        out = eb.istft_fused(S, hop, win, length, grid_cap=grid_cap)
        np.testing.assert_allclose(out, ao.istft(S, hop_length=hop, n_fft=2048, length=length), atol=1e-5)


@pytest.mark.parametrize("hop,L,B,grid_cap,Ts,variant", [
    (512, 10752, 3, 2, None, 0),    # T = 22: two 16-frame groups per clip, the second with 6 frames; clip change in a stretch
    (512, 20000, 2, 5, None, 0),    # T = 40: stretches that start inside a clip (8-frame warm-up step)
    (512, 20000, 2, 3, 48, 0),      # padded rows (whole 128-byte lines); 10 steps on 3 workgroups: a stretch starts on a second step
    (512, 20000, 2, 7, None, 0),    # 10 steps on 7 workgroups: one- and two-step stretches, odd and even starts
    (512, 20000, 2, 10, None, 0),   # one step per workgroup
    (512, 11300, 2, 1, None, 0),    # T = 23: second step with 7 frames
    (512, 12400, 2, 4, 30, 0),      # T = 25: the clip ends after a first step (one frame in the last group)
    (1024, 30000, 2, 2, None, 0),
    (256, 9000, 1, 2, None, 0),
    (512, 6000, 1, 0, None, 0),
    # the eight-round staging pass (1) and the loads issued in the staging pass (2, 3)
    (512, 20000, 2, 3, 48, 1),
    (512, 20000, 2, 7, None, 1),
    (512, 20000, 2, 5, None, 2),
    (512, 12400, 2, 4, 30, 2),
    (256, 9000, 1, 2, None, 3),
])
def test_emu_istft16(hop, L, B, grid_cap, Ts, variant):
    """kernels_istft16.h: 16-frame loads as they fall, two 8-frame overlap-add steps per load."""
    rng = np.random.default_rng(hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    S = ao.stft(y, n_fft=2048, hop_length=hop)
    win = ao.padded_window("hann", 2048, 2048)
    for length in (L, L - 700):
        out = eb.istft16(S, hop, win, length, grid_cap=grid_cap, Ts=Ts, variant=variant)
        np.testing.assert_allclose(out, ao.istft(S, hop_length=hop, n_fft=2048, length=length), atol=1e-5)
    # center=False: no trim, output longer than the frames reach (zero tail)
    # (where the window sum of squares is tiny - the first and last samples of a clip without centring - float32
    #  rounding of the frames is amplified by up to 1e8: compare where the divisor is not)
    n = L + 2048 + 100
    out = eb.istft16(S, hop, win, n, out_offset=0, grid_cap=grid_cap, Ts=Ts, variant=variant)
    ref = ao.istft(S, hop_length=hop, n_fft=2048, center=False, length=n)
    wss = np.zeros(n + 2048)
    for t in range(S.shape[-1]):
        wss[t * hop:t * hop + 2048] += win.astype(np.float64) ** 2
    ok = wss[:n] > 1e-2
    np.testing.assert_allclose(out[:, ok], ref[:, ok], atol=1e-5)
    reach = (S.shape[-1] - 1) * hop + 2048
    assert np.all(out[:, reach:] == 0.0)


@pytest.mark.parametrize("hop,L,B,grid_cap", [(256, 5376, 3, 2), (128, 4000, 2, 3), (512, 15000, 2, 2),
                                              (256, 3000, 1, 0)])
def test_emu_istft1024_fused(hop, L, B, grid_cap):
    """Fused n_fft = 1024 ISTFT kernel: carries across groups, warm-up group of a stretch that starts
    inside a clip, clip change inside a stretch, tail after the last group, shorter length."""
    rng = np.random.default_rng(hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    S = ao.stft(y, n_fft=1024, hop_length=hop)
    win = ao.padded_window("hann", 1024, 1024)
    for length in (L, L - 300):
        out = eb.istft_fused(S, hop, win, length, grid_cap=grid_cap)
        np.testing.assert_allclose(out, ao.istft(S, hop_length=hop, n_fft=1024, length=length), atol=1e-5)


def test_emu_irfft_ignores_dc_nyquist_imag():
    rng = np.random.default_rng(5)
    S = (rng.standard_normal((1, 33, 4)) + 1j * rng.standard_normal((1, 33, 4))).astype(np.complex64)
    fr = eb.irfft_frames(S, 64)
    ref = np.fft.irfft(np.transpose(S, (0, 2, 1)).astype(np.complex128), n=64, axis=-1)
    np.testing.assert_allclose(fr, ref, atol=2e-6)


def test_emu_overlap_add_direct_and_pad_frame():
    rng = np.random.default_rng(9)
    fr = rng.standard_normal((2, 9, 64)).astype(np.float32)
    w = ao.get_window("hamming", 64)
    for hop, out_len in ((16, 64 + 8 * 16), (64, 300), (1, 72), (24, 50)):
        np.testing.assert_allclose(eb.overlap_add(fr, w, hop, out_len),
                                   ao.overlap_add(fr, w, hop, out_len), rtol=1e-5, atol=1e-6)
    x = np.arange(10, dtype=np.float32)[None]
    assert eb.pad(x, 3, 2)[0].tolist() == [3, 2, 1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6]
    assert eb.pad(x, 3, 0)[0].tolist() == [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 0, 0]
    assert eb.pad(x, 3, 1)[0].tolist() == [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 9]
    sig = rng.standard_normal((2, 100)).astype(np.float32)
    np.testing.assert_array_equal(eb.frame(sig, 10, 5), ao.frame_signal(sig, 10, 5))
    with pytest.raises(ValueError, match="reflect padding requires"):
        eb.pad(x, 10, 2)
    with pytest.raises(ValueError, match="must be >= frame_length"):
        eb.frame(sig, 200, 5)


def test_emu_resample_poly_bit_exact_vs_scipy():
    """The polyphase kernel reproduces scipy.signal.resample_poly (= the reference's
    resample_poly, resample.py:279-281) bit for bit on the CPU build."""
    from conftest import load_golden
    z = load_golden("resample_scipy.npz")
    for tag, up, down in (("p13", 1, 3), ("p21", 2, 1), ("p32", 3, 2), ("p147_160", 147, 160)):
        x = z[f"{tag}_y"]
        x2 = x if x.ndim == 2 else x[None]
        want = z[f"{tag}_out"]
        want = want if want.ndim == 2 else want[None]
        np.testing.assert_array_equal(eb.resample_poly(x2, up, down), want)
    # LDS-tiled decimator (up == 1): ragged lengths, several workgroups, down = 2..5
    import scipy.signal
    rng = np.random.default_rng(1)
    # (both register blockings: two outputs per thread where that window wastes fewer positions - down 3, 4, 5, 7 -
    #  and four)
    for down, L in ((3, 4800), (2, 5001), (4, 2049), (5, 777), (3, 100), (7, 9000), (3, 20011)):
        xx = rng.standard_normal((2, L)).astype(np.float32)
        want = scipy.signal.resample_poly(xx, 1, down, axis=-1).astype(np.float32)
        np.testing.assert_array_equal(eb.resample_poly(xx, 1, down), want)
        np.testing.assert_array_equal(eb.resample_poly(xx, 1, down, two_outputs=False), want)
    # LDS-tiled interpolating cases (up > 1): several workgroups, ragged ends, large and tiny ratios
    for up, down, L in ((160, 147, 5000), (2, 1, 3001), (3, 2, 2500), (4, 3, 1025), (2, 9, 9001), (7, 5, 60), (441, 160, 400)):
        xx = rng.standard_normal((2, L)).astype(np.float32)
        np.testing.assert_array_equal(eb.resample_poly(xx, up, down),
                                      scipy.signal.resample_poly(xx, up, down, axis=-1).astype(np.float32))
    x = np.random.default_rng(0).standard_normal((2, 1000)).astype(np.float32)
    np.testing.assert_array_equal(eb.resample_linear(x, 733), ao.resample(x, 1000, 733, res_type="linear"))
    np.testing.assert_allclose(eb.resample_linear(x, 1500, scale=1.5),
                               ao.resample(x, 1000, 1500, res_type="linear", scale=True), rtol=1e-6)


def test_emu_db_dct_and_gl_projection():
    rng = np.random.default_rng(3)
    S = (rng.standard_normal((3, 40, 50)).astype(np.float32)) ** 2
    np.testing.assert_allclose(eb.to_db(S), ao.power_to_db(S), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(eb.to_db(S, coef=20.0, amin=1e-5, ref_is_max=True, top_db=None),
                               ao.amplitude_to_db(S, ref=np.max, top_db=None), rtol=1e-5, atol=1e-4)
    x = rng.standard_normal((3, 40, 50)).astype(np.float32)
    C = ao.dct_matrix(13, 40)
    got = eb.dct(x, C, 3, 40, 50).reshape(3, 13, 50)
    np.testing.assert_allclose(got, ao.dct(x, n=13, axis=1), rtol=1e-4, atol=1e-4)
    got = eb.dct(x, ao.dct_matrix(20, 50), 120, 50, 1).reshape(3, 40, 20)
    np.testing.assert_allclose(got, ao.dct(x, n=20, axis=-1), rtol=1e-4, atol=1e-4)
    # fused power_to_db (+ top_db clip against the global maximum) + DCT = the mfcc tail
    for top in (80.0, 30.0, None):
        got = eb.dct(S, C, 3, 40, 50, db=(10.0, 1e-10, 1.0, top)).reshape(3, 13, 50)
        want = ao.dct(ao.power_to_db(S, top_db=top), n=13, axis=1)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-3)
    got = eb.dct(x, ao.dct_matrix(24, 40), 3, 40, 50).reshape(3, 24, 50)       # 32-wide chunks
    np.testing.assert_allclose(got, ao.dct(x, n=24, axis=1), rtol=1e-4, atol=1e-4)
    # both offset widths are the same arithmetic; a reference level other than 1 takes the division;
    # row counts that are not a multiple of the 8-row load group take the remainder loop
    for wide in (False, True):
        for n_in, n_out in ((40, 13), (43, 13), (7, 7), (45, 24)):
            Sx = (rng.standard_normal((2, n_in, 33)).astype(np.float32)) ** 2
            Cx = ao.dct_matrix(n_out, n_in)
            np.testing.assert_array_equal(eb.dct(Sx, Cx, 2, n_in, 33, wide=wide), eb.dct(Sx, Cx, 2, n_in, 33))
            got = eb.dct(Sx, Cx, 2, n_in, 33, db=(10.0, 1e-10, 0.37, 60.0), wide=wide).reshape(2, n_out, 33)
            want = ao.dct(ao.power_to_db(Sx, ref=0.37, top_db=60.0), n=n_out, axis=1)
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-3)
    # Griffin-Lim projection: init and one momentum step, with a shorter R (zero-padded frames)
    Sm = np.abs(rng.standard_normal((2, 9, 7))).astype(np.float32)
    ang = rng.uniform(-np.pi, np.pi, Sm.shape).astype(np.float32)
    reb, tp = eb.gl_project(0, Sm, angles=ang)
    want = (Sm * np.exp(1j * ang.astype(np.float64))).astype(np.complex64)
    np.testing.assert_allclose(reb, want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tp, want, rtol=1e-5, atol=1e-6)
    R = (rng.standard_normal((2, 9, 5)) + 1j * rng.standard_normal((2, 9, 5))).astype(np.complex64)
    reb2, tp2 = eb.gl_project(1, Sm, R=R, momentum=0.99, tprev=want)
    Rp = np.pad(R, [(0, 0), (0, 0), (0, 2)])
    Rn = (Sm * np.exp(1j * np.angle(Rp).astype(np.float64))).astype(np.complex64)
    np.testing.assert_allclose(tp2, Rn, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(reb2, Rn + np.float32(0.99) * (Rn - want), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("Nx,num,B", [(1000, 500, 2), (1000, 1500, 1), (1001, 367, 1), (22050, 11025, 1),
                                      (4800, 4410, 2), (6, 4, 1)])
def test_emu_resample_fft_matches_scipy(Nx, num, B):
    """Four-step large-N FFT + SciPy's spectrum surgery == scipy.signal.resample."""
    import scipy.signal
    x = np.random.default_rng(Nx).standard_normal((B, Nx)).astype(np.float32)
    np.testing.assert_allclose(eb.resample_fft(x, num), scipy.signal.resample(x, num, axis=-1),
                               rtol=1e-4, atol=1e-5)
    assert eb.cfft_split(9001) is None and eb.cfft_split(22050) == (150, 147)


@pytest.mark.parametrize("Nx,num,B,force", [(1000, 500, 2, True), (601, 907, 1, True), (6, 4, 1, True),
                                            (10007, 5003, 1, False), (8192, 9001, 2, False), (9001, 4410, 1, False)])
def test_emu_resample_fft_chirp_matches_scipy(Nx, num, B, force):
    """Lengths with a prime factor > 4096 (and, forced, small ones): either transform as a chirp-z
    (Bluestein) convolution on the four-step engine == scipy.signal.resample."""
    import scipy.signal
    x = np.random.default_rng(Nx + num).standard_normal((B, Nx)).astype(np.float32)
    np.testing.assert_allclose(eb.resample_fft(x, num, force_chirp=force), scipy.signal.resample(x, num, axis=-1),
                               rtol=1e-4, atol=2e-5)


def test_emu_pcg64_uniform_matches_numpy_bit_for_bit():
    """Device PCG64 (128-bit LCG jump-ahead + XSL-RR) == np.random.default_rng(seed).uniform(...)"""
    for seed, n in ((42, 100000), (7, 33), (123456789, 4097)):
        want = np.random.default_rng(seed).uniform(-np.pi, np.pi, n).astype(np.float32)
        np.testing.assert_array_equal(eb.pcg64_uniform(seed, -np.pi, np.pi, n), want)
    want = np.random.default_rng(3).uniform(0.0, 1.0, 1000).astype(np.float32)
    np.testing.assert_array_equal(eb.pcg64_uniform(3, 0.0, 1.0, 1000), want)


@pytest.mark.parametrize("sr,M,hop,L,B,power,center,kw", [
    (16000, 80, 160, 5000, 3, 2.0, True, {}),          # the Whisper front end
    (16000, 80, 160, 1300, 2, 1.0, False, {}),         # ragged last group (T = 6), no centring
    (22050, 128, 100, 3000, 1, 1.5, True, {}),         # 128 filters, another hop and power
    (16000, 64, 200, 4000, 2, 2.0, True, dict(fmin=300.0, fmax=6000.0, htk=True, norm=None)),
])
def test_emu_wave400_kernel(sr, M, hop, L, B, power, center, kw):
    """kernels_frames8.h (n_fft = 400: eight frames per wave, radix-25 in registers, radix-8 across
    lanes, band contraction) on the CPU, and the compile-time LDS engine it replaces (tile_kernel)."""
    rng = np.random.default_rng(M + hop)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", 400, 400)
    fb = ao.mel_filterbank(sr, 400, M, **kw)
    R = ao.melspectrogram(y, sr=sr, n_fft=400, hop_length=hop, n_mels=M, power=power, center=center, **kw)
    A, amax = eb.melspec(y, 400, hop, win, fb, power=power, center=center, return_max=True)
    assert A.shape == R.shape
    np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)
    assert amax == A.max()
    A2 = eb.melspec(y, 400, hop, win, fb, power=power, center=center, tile_kernel=True)
    np.testing.assert_allclose(A2, R, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n_fft,sr,M,hop,L,B,power,center,kw", [
    (512, 22050, 128, 128, 4000, 2, 2.0, True, {}),        # the reference's test grid (test_stft.py:45-46)
    (512, 22050, 40, 256, 3000, 1, 1.0, False, {}),        # wide top bands (two parts), no centring
    (512, 16000, 80, 160, 2600, 2, 0.7, True, dict(htk=True)),
    (256, 8000, 32, 64, 1500, 3, 2.0, True, {}),
    (256, 16000, 64, 100, 1111, 1, 1.0, True, dict(fmin=100.0, norm=None)),
])
def test_emu_frames8_mel_kernels(n_fft, sr, M, hop, L, B, power, center, kw):
    """The even-R members of kernels_frames8.h (n_fft 512 = 32 x 8 x 2, 256 = 16 x 8 x 2): padded plane
    blocks, window in LDS for R = 32, band table built in the padded address space."""
    rng = np.random.default_rng(n_fft + M + hop)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", n_fft, n_fft)
    fb = ao.mel_filterbank(sr, n_fft, M, **kw)
    R = ao.melspectrogram(y, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=M, power=power, center=center, **kw)
    A, amax = eb.melspec(y, n_fft, hop, win, fb, power=power, center=center, return_max=True)
    assert A.shape == R.shape
    np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)
    assert amax == A.max()
    A2 = eb.melspec(y, n_fft, hop, win, fb, power=power, center=center, tile_kernel=True)   # the engine it replaces
    np.testing.assert_allclose(A2, R, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n_fft,hop,L,B,center", [
    (512, 128, 4000, 2, True), (512, 256, 2049, 1, True), (512, 512, 3000, 1, False),
    (400, 160, 3333, 2, True), (400, 77, 1000, 1, False), (256, 64, 1500, 2, True), (256, 33, 700, 1, True),
])
def test_emu_frames8_stft_kernels(n_fft, hop, L, B, center):
    """STFT members of kernels_frames8.h against the oracle (odd hops: sample pairs straddle the clip
    ends; ragged last groups; hop == n_fft)."""
    rng = np.random.default_rng(n_fft + hop)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", n_fft, n_fft)
    want = np.stack([ao.stft(y[b], n_fft=n_fft, hop_length=hop, center=center, pad_mode="constant") for b in range(B)])
    got = eb.stft(y, n_fft, hop, win, center=center, pad_mode=0)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("hop,L,B,center", [(512, 20000, 2, True), (300, 9000, 1, True), (512, 6000, 3, False)])
def test_emu_spectral_statistics_from_audio(hop, L, B, center):
    """ap_spec2048_run_kernel: transform + per-frame reductions in one kernel, against the oracle's
    feature functions (reference features.py:57-442).  Tones + noise, one silent clip region."""
    rng = np.random.default_rng(hop + L)
    t = np.arange(L) / 22050.0
    y = (0.5 * np.sin(2 * np.pi * 440.0 * t)[None] + 0.05 * rng.standard_normal((B, L))).astype(np.float32)
    y[0, : L // 3] = 0.0                                        # digital silence: all-zero frames
    win = ao.padded_window("hann", 2048, 2048)
    kw = dict(sr=22050, n_fft=2048, hop_length=hop, center=center)
    got = eb.spectral_from_audio(y, 22050, hop, win, center=center)
    for b in range(B):
        np.testing.assert_allclose(got["centroid"][b], ao.spectral_centroid(y[b], **kw)[0], rtol=2e-4, atol=1e-2)
        np.testing.assert_allclose(got["bandwidth"][b], ao.spectral_bandwidth(y[b], **kw)[0], rtol=2e-4, atol=1e-2)
        want = ao.spectral_rolloff(y[b], **kw)[0]
        off = np.abs(got["rolloff"][b] - want) / (22050 / 2048)
        assert off.max() <= 1.001 and (off > 0.5).mean() < 0.02         # at most the neighbouring bin, rarely
    flat = eb.spectral_from_audio(y, 22050, hop, win, center=center, power=2.0, want=("flatness",))["flatness"]
    for b in range(B):
        np.testing.assert_allclose(flat[b], ao.spectral_flatness(y[b], n_fft=2048, hop_length=hop, center=center)[0],
                                   rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("n_fft,B,T", [(512, 2, 19), (400, 3, 8), (256, 1, 33), (512, 1, 1)])
def test_emu_irfft8_wave_kernel(n_fft, B, T):
    """ap_irfft8_wave_kernel (kernels_frames8.h): the spectrum loaded in the forward transform's input
    layout, Hermitian merge through lane permutes, the forward machinery on the conjugate - against
    numpy.fft.irfft; imaginary parts of the DC and Nyquist bins are ignored; ragged last group."""
    rng = np.random.default_rng(n_fft + T)
    F = n_fft // 2 + 1
    S = (rng.standard_normal((B, F, T)) + 1j * rng.standard_normal((B, F, T))).astype(np.complex64)
    want = np.fft.irfft(S.astype(np.complex128), n=n_fft, axis=1).transpose(0, 2, 1)      # (B, T, n_fft)
    got = eb.irfft_frames(S, n_fft)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("n_fft,hop,L,B,grid_cap", [
    (512, 128, 9000, 3, 1), (512, 256, 5000, 2, 1), (512, 512, 7000, 1, 1), (512, 64, 3000, 2, 1),
    (400, 160, 8000, 2, 1), (400, 100, 4100, 3, 1), (256, 64, 3000, 2, 1), (256, 33, 1500, 1, 1), (512, 128, 20000, 2, 2),
])
def test_emu_istft8_fused(n_fft, hop, L, B, grid_cap):
    """Fused ISTFT of the frames8 family (ap_istft8_wave_kernel): LDS accumulation, carries along a
    stretch, warm-up group of a stretch that starts inside a clip, clip change inside a stretch, tail and
    zero fill after a clip's last group; `length` shorter and longer than the signal; no centre trim."""
    rng = np.random.default_rng(n_fft + hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    wname = "boxcar" if hop == n_fft else "hann"          # a Hann window without overlap has wss -> 0 at every seam
    S = ao.stft(y, n_fft=n_fft, hop_length=hop, window=wname)
    win = ao.padded_window(wname, n_fft, n_fft)
    for length in (L, L - 333, L + 60, L + 700):
        out = eb.istft_fused(S, hop, win, length, grid_cap=grid_cap)
        want = ao.istft(S, hop_length=hop, n_fft=n_fft, length=length, window=wname)
        # past L + 60 the window-sum-squares falls towards the 1e-8 floor and the division amplifies the
        # 1e-7 differences of any two irfft implementations: there only the zero fill is compared
        np.testing.assert_allclose(out[:, :L + 60], want[:, :L + 60], atol=1e-5)
        np.testing.assert_array_equal(out[:, L + n_fft // 2:], want[:, L + n_fft // 2:])
    out = eb.istft_fused(S, hop, win, L + n_fft, out_offset=0, grid_cap=grid_cap)      # center=False reconstruction
    want = ao.istft(S, hop_length=hop, n_fft=n_fft, center=False, length=L + n_fft, window=wname)
    np.testing.assert_allclose(out[:, n_fft // 2:L + n_fft // 2], want[:, n_fft // 2:L + n_fft // 2], atol=1e-5)


@pytest.mark.parametrize("hop,pad_mode,L,B,power", [(512, "reflect", 9000, 3, 2.0), (512, "edge", 7001, 2, 1.0),
                                                     (333, "constant", 6000, 2, 2.0), (300, "reflect", 5000, 1, 2.0)])
def test_emu_run_kernel_index_remapped_edges(hop, pad_mode, L, B, power):
    """The run kernel's third input mode: reflect / edge padding and odd hops - frames that reach over a clip end
    load through the index remap, the others through the bounds-checked loads, the register reuse between
    consecutive frames (hop 512) unchanged."""
    rng = np.random.default_rng(hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", 2048, 2048)
    fb = ao.mel_filterbank(22050, 2048, 128)
    R = ao.melspectrogram(y, sr=22050, n_fft=2048, hop_length=hop, n_mels=128, power=power, pad_mode=pad_mode)
    A, amax = eb.melspec(y, 2048, hop, win, fb, power=power, pad_mode=PM[pad_mode], return_max=True)
    np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)
    assert amax == A.max()


@pytest.mark.parametrize("n_fft,hop,pad_mode,L,B", [(512, 128, "reflect", 4000, 2), (512, 77, "constant", 3001, 2),
                                                     (400, 160, "edge", 5000, 3), (256, 64, "reflect", 1500, 1),
                                                     (400, 33, "reflect", 2000, 1)])
def test_emu_frames8_index_remapped_edges(n_fft, hop, pad_mode, L, B):
    """PADGEN instantiations of kernels_frames8.h: reflect / edge padding and centred frames at odd hops."""
    rng = np.random.default_rng(n_fft + hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", n_fft, n_fft)
    want = np.stack([ao.stft(y[b], n_fft=n_fft, hop_length=hop, pad_mode=pad_mode) for b in range(B)])
    np.testing.assert_allclose(eb.stft(y, n_fft, hop, win, pad_mode=PM[pad_mode]), want, rtol=1e-4, atol=1e-4)
    fb = ao.mel_filterbank(16000, n_fft, 40)
    R = ao.melspectrogram(y, sr=16000, n_fft=n_fft, hop_length=hop, n_mels=40, pad_mode=pad_mode)
    A, amax = eb.melspec(y, n_fft, hop, win, fb, pad_mode=PM[pad_mode], return_max=True)
    np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)
    assert amax == A.max()                 # only the fused kernels hand the maximum back


def test_zz_no_kernel_wrote_past_its_dynamic_lds():
    """Runs last in this file: every emulated launch above poisoned the LDS beyond the lds_bytes its host code
    computed and checked it afterwards (tests/emu/emu_shim.h)."""
    assert eb.lds_overruns() == 0
    assert eb.lib().emu_lds_guard_selftest() == 1          # the guard does see a write past the limit
    assert eb.lds_overruns() == 0


@pytest.mark.parametrize("hop,pad_mode,L,B", [(256, "reflect", 6000, 2), (256, "edge", 3001, 3), (255, "constant", 4000, 2),
                                              (300, "reflect", 5000, 1)])
def test_emu_mel1024_index_remapped_edges(hop, pad_mode, L, B):
    """PADGEN instantiation of the n_fft = 1024 mel kernel: reflect / edge padding, centred frames at odd hops."""
    rng = np.random.default_rng(hop + L)
    y = rng.standard_normal((B, L)).astype(np.float32)
    win = ao.padded_window("hann", 1024, 1024)
    fb = ao.mel_filterbank(22050, 1024, 80)
    R = ao.melspectrogram(y, sr=22050, n_fft=1024, hop_length=hop, n_mels=80, pad_mode=pad_mode)
    A, amax = eb.melspec(y, 1024, hop, win, fb, pad_mode=PM[pad_mode], return_max=True)
    np.testing.assert_allclose(A, R, rtol=1e-4, atol=1e-4)
    assert amax == A.max()                 # only the fused kernels hand the maximum back
