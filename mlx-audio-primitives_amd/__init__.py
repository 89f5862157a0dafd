"""MI355X-native audio-DSP primitives with the librosa-compatible API of
zkeown/mlx-audio-primitives for its feature-extraction hot path.

Tensors are torch tensors in HBM; every device op is a hand-written gfx950 HIP
kernel behind the C ABI in include/audioprims.h.  There is no CPU fallback.
"""

from ._extension import HAS_HIP_EXT, _ext
from ._validation import validate_non_negative, validate_positive, validate_range
from .convert import amplitude_to_db, db_to_amplitude, db_to_power, power_to_db
from .features import (spectral_bandwidth, spectral_centroid, spectral_contrast, spectral_features, spectral_flatness,
                       spectral_rolloff, zero_crossing_rate)
from .filterbanks import bark_filterbank, bark_to_hz, hz_to_bark, linear_filterbank
from .framing import deemphasis, frame, preemphasis, rms
from .griffinlim import griffinlim, griffinlim_iter
from .mel import filterbank_spectrogram, hz_to_mel, mel_filterbank, mel_to_hz, melspectrogram, pcm16_to_float
from .mfcc import dct, delta, mfcc
from .pitch import autocorrelation, periodicity, pitch_detect_acf
from .resample import resample, resample_poly
from .stft import check_nola, istft, magnitude, phase, set_spectrum_layout, stft, stft_padded_rows
from .streaming import StreamingSTFT
from .windows import get_window

__version__ = "0.1.0"

__all__ = [
    "__version__",
    "HAS_HIP_EXT", "_ext",
    "stft", "istft", "magnitude", "phase", "check_nola",
    "get_window",
    "hz_to_mel", "mel_to_hz", "mel_filterbank", "melspectrogram",
    "griffinlim", "griffinlim_iter", "resample", "resample_poly",
    "set_spectrum_layout", "stft_padded_rows",
    "spectral_centroid", "spectral_bandwidth", "spectral_rolloff", "spectral_flatness", "spectral_contrast", "spectral_features",
    "StreamingSTFT", "autocorrelation", "pitch_detect_acf", "periodicity", "pcm16_to_float", "hz_to_bark", "bark_to_hz", "bark_filterbank", "linear_filterbank", "filterbank_spectrogram",
    "zero_crossing_rate", "frame", "rms", "preemphasis", "deemphasis", "delta",
    "mfcc", "dct", "power_to_db", "db_to_power", "amplitude_to_db", "db_to_amplitude",
    "validate_positive", "validate_non_negative", "validate_range",
]
