"""Generate the committed golden fixtures under tests/golden/.

Run in the BUILD container:  python tests/golden/make_golden.py

None of the numbers below come from oracle/ or from our HIP code: every array
is produced by a third-party implementation that the reference's own tests use
as *their* oracle and that happens to be installed in this image, or is a
closed-form known answer copied (as data) from the reference's tests:

  torch.stft / torch.istft     tests/test_torchaudio_crossval.py:26-106, :278-322
  scipy.signal.get_window      tests/test_windows.py:24-48
  scipy.fft.dct                tests/test_mfcc.py:190-232
  scipy.signal.resample        resample.py:97,123  (the reference's own call)
  scipy.signal.resample_poly   resample.py:279-281 (the reference's own call)
  transformers.audio_utils.mel_filter_bank (slaney/slaney, htk)  — librosa-compatible
                               stand-in for librosa.filters.mel, tests/test_mel.py:74-136
  pad / HTK known answers      tests/test_cpp_extension.py:354-380, :525-546

The reference itself cannot run here (ModuleNotFoundError: mlx) and
librosa is not installed, so there is no reference-generated vector in this set.
"""

from __future__ import annotations

import os

import numpy as np
import scipy.fft
import scipy.signal
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
F32 = np.float32


def _signal(n, seed=42):
    return np.random.default_rng(seed).standard_normal(n).astype(F32)


def make_stft():
    out = {}
    y = _signal(4096)
    yb = np.stack([_signal(3000, seed=s) for s in (1, 2, 3)])
    cases = [
        # name, signal, n_fft, hop, win_length, center, pad_mode, window
        ("c0", y, 512, 128, 512, True, "constant", "hann"),
        ("c1", y, 512, 128, 512, False, "constant", "hann"),
        ("c2", y, 1024, 256, 1024, True, "reflect", "hann"),
        ("c3", y, 400, 160, 400, True, "constant", "hann"),
        ("c4", y, 512, 200, 300, True, "constant", "hamming"),
        ("c5", yb, 256, 64, 256, True, "constant", "hann"),
        ("c6", y, 2048, 512, 2048, True, "constant", "hann"),
        ("c7", y, 64, 16, 64, True, "reflect", "blackman"),
    ]
    torch_windows = {
        "hann": torch.hann_window,
        "hamming": torch.hamming_window,
        "blackman": torch.blackman_window,
    }
    meta = []
    for name, sig, n_fft, hop, wl, center, pad_mode, window in cases:
        w = torch_windows[window](wl, periodic=True, dtype=torch.float64)
        t = torch.from_numpy(sig.astype(np.float64))
        S = torch.stft(t, n_fft=n_fft, hop_length=hop, win_length=wl, window=w,
                       center=center, pad_mode=pad_mode, return_complex=True)
        out[f"{name}_y"] = sig
        out[f"{name}_S"] = S.numpy().astype(np.complex64)
        # torch.istft round trip of torch's own spectrum (center only)
        if center:
            yi = torch.istft(S, n_fft=n_fft, hop_length=hop, win_length=wl, window=w,
                             center=True, length=sig.shape[-1])
            out[f"{name}_istft"] = yi.numpy().astype(F32)
        meta.append(f"{name},{n_fft},{hop},{wl},{int(center)},{pad_mode},{window}")
    out["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, "stft_torch.npz"), **out)


def make_windows():
    out = {}
    for name in ("hann", "hamming", "blackman", "bartlett", "boxcar"):
        for n in (4, 16, 255, 400, 512, 2048):
            for periodic in (True, False):
                w = scipy.signal.get_window(name, n, fftbins=periodic)
                out[f"{name}_{n}_{int(periodic)}"] = w.astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "windows_scipy.npz"), **out)


def make_mel():
    from transformers.audio_utils import mel_filter_bank

    out = {}
    meta = []
    cases = [
        (22050, 2048, 128, 0.0, 11025.0, "slaney", "slaney"),
        (22050, 2048, 40, 0.0, 11025.0, "slaney", "slaney"),
        (22050, 1024, 64, 0.0, 11025.0, "slaney", "slaney"),
        (16000, 400, 80, 0.0, 8000.0, "slaney", "slaney"),
        (22050, 2048, 80, 300.0, 8000.0, "slaney", "slaney"),
        (22050, 2048, 128, 0.0, 11025.0, None, "slaney"),
        (22050, 2048, 128, 0.0, 11025.0, "slaney", "htk"),
    ]
    for i, (sr, n_fft, n_mels, fmin, fmax, norm, scale) in enumerate(cases):
        fb = mel_filter_bank(
            num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels,
            min_frequency=fmin, max_frequency=fmax, sampling_rate=sr,
            norm=norm, mel_scale=scale,
        )  # (F, M) float64
        out[f"fb{i}"] = fb.T.astype(np.float64)
        meta.append(f"fb{i},{sr},{n_fft},{n_mels},{fmin},{fmax},{norm},{scale}")
    out["meta"] = np.array(meta)
    # HTK closed forms (tests/test_cpp_extension.py:354-380): mel = 2595*log10(1+f/700)
    out["htk_hz"] = np.array([0.0, 100.0, 1000.0, 8000.0])
    out["htk_mel"] = 2595.0 * np.log10(1.0 + out["htk_hz"] / 700.0)
    # Slaney scale anchor points: 0 Hz -> 0 mel, 1000 Hz -> 15 mel, 6400 Hz -> 42 mel
    out["slaney_hz"] = np.array([0.0, 200.0, 1000.0, 6400.0])
    out["slaney_mel"] = np.array([0.0, 3.0, 15.0, 42.0])
    np.savez_compressed(os.path.join(HERE, "mel_filters.npz"), **out)


def make_dct():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((5, 128)).astype(F32)
    out = {"x": x}
    out["ortho_full"] = scipy.fft.dct(x.astype(np.float64), type=2, norm="ortho", axis=-1)
    out["ortho_13"] = out["ortho_full"][:, :13]
    # scipy's un-normalised DCT-II is 2*sum(...); the reference's norm=None basis is
    # the bare cosine sum (mfcc.py:54-60), i.e. half of scipy's.
    out["none_full_half_scipy"] = 0.5 * scipy.fft.dct(x.astype(np.float64), type=2, norm=None, axis=-1)
    np.savez_compressed(os.path.join(HERE, "dct_scipy.npz"), **out)


def make_resample():
    out = {}
    y = _signal(4800, seed=11)
    yb = np.stack([_signal(1500, seed=s) for s in (21, 22)])
    for tag, sig, up, down in (("p13", y, 1, 3), ("p21", y, 2, 1), ("p32", yb, 3, 2),
                               ("p147_160", yb, 147, 160)):
        out[f"{tag}_y"] = sig
        out[f"{tag}_out"] = scipy.signal.resample_poly(sig, up, down, axis=-1,
                                                       padtype="constant").astype(F32)
    for tag, sig, num in (("f_half", y, 2400), ("f_up", y[:1000], 1500), ("f_odd", y[:1001], 367),
                          ("f_b", yb, 1102)):
        out[f"{tag}_y"] = sig
        out[f"{tag}_out"] = scipy.signal.resample(sig, num, axis=-1).astype(F32)
    np.savez_compressed(os.path.join(HERE, "resample_scipy.npz"), **out)


def make_kats():
    """Known answers copied as data from tests/test_cpp_extension.py:525-546."""
    out = {
        "pad_in": np.arange(10, dtype=F32)[None, :],
        "pad_reflect_3": np.array([[3, 2, 1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6]], dtype=F32),
        "pad_constant_3": np.array([[0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 0, 0]], dtype=F32),
        "pad_edge_3": np.array([[0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 9]], dtype=F32),
    }
    np.savez_compressed(os.path.join(HERE, "kats.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(1)
    make_stft()
    make_windows()
    make_mel()
    make_dct()
    make_resample()
    make_kats()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
