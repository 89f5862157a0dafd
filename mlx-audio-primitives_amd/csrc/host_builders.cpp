// Host-side float64 builders behind the C ABI: windows, mel scale / filterbank, DCT
// basis, twiddle table.  These mirror the parts of the reference extension that are
// deliberately run on the CPU stream in double precision
// (windows.cpp:179-228, mel_filterbank.cpp:70-239, dct.cpp:24-101).
#include <algorithm>
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

namespace {
struct MelPart { int row, g0, ng, q0, slot; };

// spans + parts of a dense filterbank; returns total quads
void mel_analyse(const float *fb, int M, int F, std::vector<int> &lo, std::vector<int> &len,
                 std::vector<MelPart> &parts) {
    lo.assign(M, 0);
    len.assign(M, 0);
    parts.clear();
    for (int m = 0; m < M; ++m) {
        int a = 0, b = 0;
        bool any = false;
        for (int k = 0; k < F; ++k)
            if (fb[(size_t)m * F + k] != 0.0f) { if (!any) a = k; b = k + 1; any = true; }
        if (!any) continue;
        lo[m] = a;
        len[m] = b - a;
        const int gfirst = a >> 2, glast = (b - 1) >> 2;
        for (int g = gfirst; g <= glast; g += 4) {
            MelPart p;
            p.row = m;
            p.g0 = g;
            p.ng = (glast - g + 1) < 4 ? (glast - g + 1) : 4;
            p.q0 = 0;
            p.slot = (int)parts.size();       // row-major position: a row's parts are adjacent
            parts.push_back(p);
        }
    }
    // longest parts first: threads of one wave then run loops of (nearly) equal length
    std::stable_sort(parts.begin(), parts.end(),
                     [](const MelPart &x, const MelPart &y) { return x.ng > y.ng; });
    int q = 0;
    for (auto &p : parts) { p.q0 = q; q += p.ng; }
}

// Wave layout of the same parts for the wave-per-frame kernels (kernels_wave.h): passes of 64
// parts, lane l of pass p owns entry 64 p + l.  Within a pass the parts are dealt to the four
// 16-lane groups that one ds_read_b128 serves together so that the 16-byte |X|^p groups they
// read fall into different banks (first group index distinct modulo 16) wherever the plan
// allows it; the weight quads are stored lane-interleaved (group i of entry e at
// qw(e) + 64 i), which makes every weight read of a pass one contiguous 1 KiB row.
struct MelWaveLayout {
    // 64 * passes entries; an entry computes two half sums: A over weight rows 0-1 (|X|^p groups
    // gA, gA + 1) and B over rows 2-3 (groups gB, gB + 1).  A part of 3-4 groups takes a whole
    // entry (gB = gA + 2, both halves go to its slot); two parts of <= 2 groups share one.
    std::vector<int> partA, partB;     // index into parts, or -1
    int n_quads = 0;
};

const int kB128Groups[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

void mel_wave_layout(const std::vector<MelPart> &parts, MelWaveLayout &L) {
    // parts are sorted by length (longest first): whole entries first, then the pairs
    std::vector<std::pair<int, int>> ent;
    size_t i = 0;
    for (; i < parts.size() && parts[i].ng > 2; ++i) ent.push_back({(int)i, -1});
    for (; i < parts.size(); i += 2) ent.push_back({(int)i, i + 1 < parts.size() ? (int)i + 1 : -1});
    const int n = (int)ent.size();
    const int passes = (n + 63) / 64;
    L.partA.assign((size_t)passes * 64, -1);
    L.partB.assign((size_t)passes * 64, -1);
    for (int ps = 0; ps < passes; ++ps) {
        const int first = ps * 64, last = first + 64 < n ? first + 64 : n;
        // deal the entries of the pass to the four 16-lane groups of a ds_read_b128 so that their
        // first |X|^p groups fall into different banks (distinct modulo 16) where the plan allows
        std::vector<int> bucket[16];
        for (int e = first; e < last; ++e) bucket[parts[ent[e].first].g0 & 15].push_back(e);
        int order[16];
        for (int r = 0; r < 16; ++r) order[r] = r;
        std::stable_sort(order, order + 16, [&](int a, int b) { return bucket[a].size() > bucket[b].size(); });
        std::vector<int> grp[4];
        int has[4][16] = {};
        for (int oi = 0; oi < 16; ++oi) {
            const int r = order[oi];
            for (int idx : bucket[r]) {
                int best = -1;
                for (int g = 0; g < 4; ++g) {
                    if (grp[g].size() >= 16) continue;
                    if (best < 0 || has[g][r] < has[best][r] ||
                        (has[g][r] == has[best][r] && grp[g].size() < grp[best].size()))
                        best = g;
                }
                grp[best].push_back(idx);
                has[best][r]++;
            }
        }
        for (int g = 0; g < 4; ++g)
            for (size_t j = 0; j < grp[g].size(); ++j) {
                const size_t slot = (size_t)ps * 64 + kB128Groups[g][j];
                L.partA[slot] = ent[grp[g][j]].first;
                L.partB[slot] = ent[grp[g][j]].second;
            }
    }
    // every pass owns four 64-quad weight rows; entry e = 64 p + l reads row i at 256 p + 64 i + l
    L.n_quads = passes * 256;
}
}  // namespace

int64_t ap_mel_plan_words(const float *fb, int n_mels, int n_bins) {
    if (!fb || n_mels <= 0 || n_bins <= 0) return 0;
    std::vector<int> lo, len;
    std::vector<MelPart> parts;
    mel_analyse(fb, n_mels, n_bins, lo, len, parts);
    int64_t quads = 0;
    for (auto &p : parts) quads += p.ng;
    MelWaveLayout L;
    mel_wave_layout(parts, L);
    return 2 * (int64_t)n_mels + 4 * (int64_t)parts.size() + 4 * quads + (int64_t)n_mels + 1 + 8 +
           4 * (int64_t)L.partA.size() + 4 * (int64_t)L.n_quads + 8;
}

int ap_mel_plan_host(const float *fb, int n_mels, int n_bins, int32_t *plan, int32_t *desc) {
    if (!fb || !plan || !desc || n_mels <= 0 || n_bins <= 0) {
        ap_set_error("mel plan: bad arguments");
        return AP_ERR_INVALID;
    }
    const int M = n_mels, F = n_bins;
    std::vector<int> lo, len;
    std::vector<MelPart> parts;
    mel_analyse(fb, M, F, lo, len, parts);
    int64_t quads = 0;
    for (auto &p : parts) quads += p.ng;
    const int64_t words = ap_mel_plan_words(fb, M, F);
    std::memset(plan, 0, sizeof(int32_t) * (size_t)words);
    std::memset(desc, 0, sizeof(int32_t) * AP_PLAN_DESC_INTS);
    int64_t off = 0;
    const int64_t off_lo = off; off += M;
    const int64_t off_len = off; off += M;
    off += off & 3 ? 4 - (off & 3) : 0;                       // 16-byte align the int4 / float4 tables
    const int64_t off_parts = off; off += 4 * (int64_t)parts.size();
    const int64_t off_quads = off; off += 4 * quads;
    const int64_t off_rs = off; off += (int64_t)M + 1;
    off += off & 3 ? 4 - (off & 3) : 0;
    MelWaveLayout WL;
    mel_wave_layout(parts, WL);
    const int64_t off_wparts = off; off += 4 * (int64_t)WL.partA.size();
    const int64_t off_wquads = off; off += 4 * (int64_t)WL.n_quads;
    for (int m = 0; m < M; ++m) { plan[off_lo + m] = lo[m]; plan[off_len + m] = len[m]; }
    bool parts_ok = true;
    // rowstart[m] .. rowstart[m+1]: the row-major slots that hold row m's partial sums
    {
        std::vector<int> cnt(M, 0);
        for (auto &p : parts) cnt[p.row]++;
        int acc = 0;
        for (int m = 0; m < M; ++m) { plan[off_rs + m] = acc; acc += cnt[m]; }
        plan[off_rs + M] = acc;
    }
    float *wq = reinterpret_cast<float *>(plan + off_quads);
    for (size_t i = 0; i < parts.size(); ++i) {
        const MelPart &p = parts[i];
        plan[off_parts + 4 * i + 0] = p.slot;
        plan[off_parts + 4 * i + 1] = p.g0;
        plan[off_parts + 4 * i + 2] = p.ng;
        plan[off_parts + 4 * i + 3] = p.q0;
        for (int g = 0; g < p.ng; ++g)
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * (p.g0 + g) + e;
                wq[4 * (size_t)(p.q0 + g) + e] = k < F ? fb[(size_t)p.row * F + k] : 0.0f;
            }
    }
    float *wwq = reinterpret_cast<float *>(plan + off_wquads);
    int max_row_parts = 0;
    for (int m = 0; m < M; ++m)
        if (plan[off_rs + m + 1] - plan[off_rs + m] > max_row_parts)
            max_row_parts = plan[off_rs + m + 1] - plan[off_rs + m];
    const int32_t dump = (int32_t)parts.size();              // slot past the last one: sums nobody reads
    auto put_rows = [&](size_t e, int row0, const MelPart &p, int g_first, int g_count) {
        const size_t col = (e / 64) * 256 + (e % 64);
        for (int g = 0; g < g_count; ++g)
            for (int c = 0; c < 4; ++c) {
                const int k = 4 * (p.g0 + g_first + g) + c;
                wwq[4 * (col + 64 * (size_t)(row0 + g)) + c] =
                    (g_first + g < p.ng && k < F) ? fb[(size_t)p.row * F + k] : 0.0f;
            }
    };
    for (size_t e = 0; e < WL.partA.size(); ++e) {
        int32_t *d = plan + off_wparts + 4 * e;                // slot A, group A, slot B, group B
        d[0] = dump; d[1] = 0; d[2] = dump; d[3] = 0;          // idle lane: zero weights, group 0
        if (WL.partA[e] < 0) continue;
        const MelPart &pa = parts[WL.partA[e]];
        d[0] = pa.slot;
        d[1] = pa.g0;
        put_rows(e, 0, pa, 0, 2);
        if (WL.partB[e] >= 0) {                                // a second short part in rows 2-3
            const MelPart &pb = parts[WL.partB[e]];
            d[2] = pb.slot;
            d[3] = pb.g0;
            put_rows(e, 2, pb, 0, 2);
        } else {                                               // rows 2-3 continue part A
            d[2] = pa.ng > 2 ? -1 : dump;                      // -1: half B belongs to slot A
            d[3] = pa.g0 + 2;
            put_rows(e, 2, pa, 2, 2);
        }
    }
    desc[0] = AP_PLAN_BANDED | (parts_ok ? AP_PLAN_PARTS : 0);
    desc[1] = M;
    desc[2] = F;
    desc[3] = (int32_t)words;
    desc[4] = (int32_t)off_lo;
    desc[5] = (int32_t)off_len;
    desc[6] = (int32_t)off_parts;
    desc[7] = (int32_t)parts.size();
    desc[8] = (int32_t)off_quads;
    desc[9] = (int32_t)quads;
    desc[10] = (int32_t)off_rs;
    desc[11] = (int32_t)off_wparts;
    desc[12] = (int32_t)WL.partA.size();
    desc[13] = (int32_t)off_wquads;
    desc[14] = (int32_t)WL.n_quads;
    desc[15] = max_row_parts;
    return AP_OK;
}

namespace {
double bessel_i0(double x) {
    // power series; |x| <= ~10 here (beta = 5): converges to double precision in < 40 terms
    const double q = 0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}
}  // namespace

int ap_resample_poly_ntaps(int up, int down) {
    if (up < 1 || down < 1) return 0;
    const int max_rate = up > down ? up : down;
    const int half_len = 10 * max_rate;
    return 2 * half_len + 1 + (down - half_len % down);
}

int ap_resample_poly_taps_host(int up, int down, float *out, int *n_pre_remove_out) {
    if (up < 1 || down < 1 || !out || !n_pre_remove_out) {
        ap_set_error("resample_poly taps: bad arguments");
        return AP_ERR_INVALID;
    }
    const int max_rate = up > down ? up : down;
    const int half_len = 10 * max_rate;
    const int numtaps = 2 * half_len + 1;
    const int n_pre_pad = down - half_len % down;
    const double fc = 1.0 / max_rate;             // cutoff relative to Nyquist
    const double alpha = 0.5 * (numtaps - 1);
    const double beta = 5.0;
    std::vector<double> h(numtaps);
    double sum = 0.0;
    for (int n = 0; n < numtaps; ++n) {
        const double m = n - alpha;
        const double a = kPi * (fc * m);          // np.sinc(fc * m): pi * x - the rounding of x decides the "zeros"
        const double sinc = (m == 0.0) ? 1.0 : std::sin(a) / a;
        const double r = (n - alpha) / alpha;
        const double arg = 1.0 - r * r;
        const double win = bessel_i0(beta * std::sqrt(arg > 0.0 ? arg : 0.0)) / bessel_i0(beta);
        h[n] = fc * sinc * win;
        sum += h[n];
    }
    for (int i = 0; i < n_pre_pad; ++i) out[i] = 0.0f;
    for (int n = 0; n < numtaps; ++n) {
        float v = (float)(h[n] / sum);            // firwin(...).astype(float32)
        v *= (float)up;                           // h *= up in float32
        out[n_pre_pad + n] = v;
    }
    *n_pre_remove_out = (half_len + n_pre_pad) / down;
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
