// Host-side float64 builders behind the C ABI: windows, mel scale / filterbank, DCT
// basis, twiddle table.  These mirror the parts of the reference extension that are
// deliberately run on the CPU stream in double precision
// (windows.cpp:179-228, mel_filterbank.cpp:70-239, dct.cpp:24-101).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/audioprims.h"
#include "ap_common.h"

void ap_set_error(const char *msg);   // audioprims.hip

namespace {

const double kPi = 3.14159265358979323846;

// Slaney scale constants (mel.py:24-28, mel_filterbank.cpp anonymous namespace)
const double F_MIN = 0.0;
const double F_SP = 200.0 / 3.0;
const double MIN_LOG_HZ = 1000.0;
const double MIN_LOG_MEL = (MIN_LOG_HZ - F_MIN) / F_SP;
const double LOGSTEP = std::log(6.4) / 27.0;

double hz_to_mel1(double f, int htk) {
    if (htk) return 2595.0 * std::log10(1.0 + f / 700.0);
    if (f < MIN_LOG_HZ) return (f - F_MIN) / F_SP;
    return MIN_LOG_MEL + std::log(f / MIN_LOG_HZ) / LOGSTEP;
}

double mel_to_hz1(double m, int htk) {
    if (htk) return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
    if (m < MIN_LOG_MEL) return F_MIN + F_SP * m;
    return MIN_LOG_HZ * std::exp(LOGSTEP * (m - MIN_LOG_MEL));
}

// numpy.linspace(start, stop, num) semantics: start + i*step, last point exact.
void linspace(double start, double stop, int num, std::vector<double> &out) {
    out.resize(num);
    if (num == 1) { out[0] = start; return; }
    const double step = (stop - start) / (num - 1);
    for (int i = 0; i < num; ++i) out[i] = start + i * step;
    out[num - 1] = stop;
}

}  // namespace

extern "C" {

int ap_generate_window_host(int kind, int length, int periodic, float *out_host) {
    if (length <= 0) { ap_set_error("Window length must be positive"); return AP_ERR_INVALID; }
    if (kind < AP_WIN_HANN || kind > AP_WIN_RECTANGULAR) {
        ap_set_error("Unknown window type. Supported: hann, hamming, blackman, bartlett, rectangular");
        return AP_ERR_INVALID;
    }
    if (!out_host) { ap_set_error("out_host is NULL"); return AP_ERR_INVALID; }
    const int n = periodic ? length + 1 : length;
    std::vector<float> w(n);
    if (kind == AP_WIN_RECTANGULAR || n <= 1) {
        for (int i = 0; i < n; ++i) w[i] = 1.0f;
    } else {
        const double denom = (double)(n - 1);
        for (int k = 0; k < n; ++k) {
            double v;
            switch (kind) {
                case AP_WIN_HANN: v = 0.5 - 0.5 * std::cos(2.0 * kPi * k / denom); break;
                case AP_WIN_HAMMING: v = 0.54 - 0.46 * std::cos(2.0 * kPi * k / denom); break;
                case AP_WIN_BLACKMAN:
                    v = 0.42 - 0.5 * std::cos(2.0 * kPi * k / denom) + 0.08 * std::cos(4.0 * kPi * k / denom);
                    if (v < 0.0) v = 0.0;
                    break;
                default: v = 1.0 - std::fabs(2.0 * k / denom - 1.0); break;   // bartlett
            }
            w[k] = (float)v;
        }
        // exact symmetry: average with the reverse in float32 (windows.cpp:73-78)
        for (int k = 0; k < n / 2; ++k) {
            const float s = (w[k] + w[n - 1 - k]) / 2.0f;
            w[k] = s;
            w[n - 1 - k] = s;
        }
    }
    std::memcpy(out_host, w.data(), sizeof(float) * (size_t)length);
    return AP_OK;
}

int ap_hz_to_mel_host(const double *hz, int64_t n, int htk, double *out) {
    if (n < 0 || (n > 0 && (!hz || !out))) { ap_set_error("bad array"); return AP_ERR_INVALID; }
    for (int64_t i = 0; i < n; ++i) out[i] = hz_to_mel1(hz[i], htk);
    return AP_OK;
}

int ap_mel_to_hz_host(const double *mel, int64_t n, int htk, double *out) {
    if (n < 0 || (n > 0 && (!mel || !out))) { ap_set_error("bad array"); return AP_ERR_INVALID; }
    for (int64_t i = 0; i < n; ++i) out[i] = mel_to_hz1(mel[i], htk);
    return AP_OK;
}

int ap_mel_filterbank_host(int sr, int n_fft, int n_mels, double fmin, double fmax, int htk,
                           int norm_slaney, float *out_host) {
    if (sr <= 0) { ap_set_error("Sample rate (sr) must be positive"); return AP_ERR_INVALID; }
    if (n_fft <= 0) { ap_set_error("n_fft must be positive"); return AP_ERR_INVALID; }
    if (n_mels <= 0) { ap_set_error("n_mels must be positive"); return AP_ERR_INVALID; }
    if (fmin < 0) { ap_set_error("fmin must be non-negative"); return AP_ERR_INVALID; }
    if (fmax < 0) fmax = sr / 2.0;
    if (fmin >= fmax) { ap_set_error("fmin must be less than fmax"); return AP_ERR_INVALID; }
    if (fmax > sr / 2.0) {
        ap_set_error("fmax cannot exceed Nyquist frequency (sr / 2)");
        return AP_ERR_INVALID;
    }
    if (!out_host) { ap_set_error("out_host is NULL"); return AP_ERR_INVALID; }
    const int F = 1 + n_fft / 2;
    std::vector<double> fftfreqs, melpts;
    linspace(0.0, sr / 2.0, F, fftfreqs);
    linspace(hz_to_mel1(fmin, htk), hz_to_mel1(fmax, htk), n_mels + 2, melpts);
    std::vector<double> mel_f(n_mels + 2);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz1(melpts[i], htk);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i];
        const double fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = norm_slaney ? 2.0 / (mel_f[i + 2] - mel_f[i]) : 1.0;
        for (int k = 0; k < F; ++k) {
            const double lower = -(mel_f[i] - fftfreqs[k]) / fd0;
            const double upper = (mel_f[i + 2] - fftfreqs[k]) / fd1;
            double v = lower < upper ? lower : upper;
            if (!(v > 0.0)) v = 0.0;
            out_host[(size_t)i * F + k] = (float)(v * enorm);
        }
    }
    return AP_OK;
}

int ap_dct_matrix_host(int n_out, int n_in, int ortho, float *out_host) {
    if (n_out <= 0 || n_in <= 0) { ap_set_error("DCT sizes must be positive"); return AP_ERR_INVALID; }
    if (!out_host) { ap_set_error("out_host is NULL"); return AP_ERR_INVALID; }
    // float32 arithmetic like the reference's native builder (dct.cpp:59-89)
    const float pi = (float)kPi;
    const float s0 = 1.0f / std::sqrt((float)n_in);
    const float s1 = std::sqrt(2.0f / (float)n_in);
    for (int k = 0; k < n_out; ++k) {
        for (int i = 0; i < n_in; ++i) {
            // (pi*k) * ((2i+1)/(2*n_in)), each step rounded to float32 (dct.cpp:57-62)
            const float a = (pi * (float)k) * (((float)i * 2.0f + 1.0f) / (2.0f * (float)n_in));
            float v = std::cos(a);
            if (ortho) v *= (k == 0) ? s0 : s1;
            out_host[(size_t)k * n_in + i] = v;
        }
    }
    return AP_OK;
}

int ap_twiddle_table_host(int n_fft, float *out_host) {
    if (n_fft <= 0 || !out_host) { ap_set_error("bad twiddle request"); return AP_ERR_INVALID; }
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int j = 0; j < n_fft; ++j) {
        long double c, s;
        // exact values at the quadrant points
        if ((4LL * j) % n_fft == 0) {
            const int q = (int)((4LL * j) / n_fft);
            c = (q == 0) ? 1.0L : (q == 2 ? -1.0L : 0.0L);
            s = (q == 1) ? 1.0L : (q == 3 ? -1.0L : 0.0L);
        } else {
            const long double a = two_pi * (long double)j / (long double)n_fft;
            c = cosl(a);
            s = sinl(a);
        }
        out_host[2 * j] = (float)c;
        out_host[2 * j + 1] = (float)s;
    }
    return AP_OK;
}

int ap_fft_supported(int n_fft) {
    ApFftPlan pl;
    ApTile tl;
    if (ap_make_plan(n_fft, &pl) != 0) return 0;
    if (ap_make_tile(&pl, 1, &tl) != 0) return 0;
    return 1;
}

}  // extern "C"
