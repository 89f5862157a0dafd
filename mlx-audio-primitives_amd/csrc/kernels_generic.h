// Generic (any n_fft) kernels of the hot path: fused pad+frame+window+rFFT with a
// complex or mel epilogue, irfft of frames, overlap-add, and the small copy
// primitives the reference extension exposes.  One 256-thread workgroup handles a
// tile of G consecutive frames of one clip, entirely in LDS.
//
// Replaces: pad_signal.metal:10-121, frame_signal.metal:10-36 (never materialised
// here), `frames * win` + mx.fft.rfft (stft.py:129-130), the transpose
// (stft.py:216), mx.abs/mx.power/mx.matmul (mel.py:321-350), mx.fft.irfft
// (stft.py:295) and overlap_add.metal:16-55.
#pragma once
#include "fft_lds.h"

#ifndef AP_PAD_CONSTANT
#define AP_PAD_CONSTANT 0
#define AP_PAD_EDGE 1
#define AP_PAD_REFLECT 2
#endif

// Sample p of the virtually padded clip (p relative to the unpadded clip, may be
// negative or >= L).  Index remaps follow pad_signal.metal:30-38 / stft.py:441-468.
// Branch-free on purpose (selects + one load): per-sample early returns made every call a
// divergent region and the edge frames of the wave kernels ten times slower than the others.
AP_DEV float ap_load_padded(const float *yb, int64_t L, int64_t p, int mode) {
    const bool out = p < 0 || p >= L;
    int64_t q = p < 0 ? -p : 2 * (L - 1) - p;   // reflect, edge sample not repeated
    if (mode != AP_PAD_REFLECT) q = p;           // wave-uniform
    q = q < 0 ? 0 : (q >= L ? L - 1 : q);       // edge: clamp; reflect: host validates pad <= L-1, guard only
    if (!out) q = p;
    const float v = yb[q];
    return (out && mode == AP_PAD_CONSTANT) ? 0.0f : v;
}

AP_DEV float ap_pow_mag(float re, float im, float power) {
    const float p2 = re * re + im * im;
    if (power == 2.0f) return p2;
    const float mag = sqrtf(p2);
    if (power == 1.0f) return mag;
    return powf(mag, power);
}

extern __shared__ __attribute__((aligned(16))) char ap_smem[];

// EPI = 0: complex spectrum (B,F,T);  EPI = 1: mel (B,M,T)
template <int EPI>
__global__ void __launch_bounds__(AP_BLOCK) ap_stft_generic_kernel(ApStftParams P) {
    const ApFftPlan &pl = P.plan;
    const int G = P.tile.G, fstride = P.tile.fstride, nc = pl.nc, n = pl.n;
    ap_float2 *bufA = reinterpret_cast<ap_float2 *>(ap_smem);
    ap_float2 *bufB = bufA + (size_t)G * fstride;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int64_t bid = blockIdx.x;
    const int64_t b = bid / P.tiles_per_clip;
    const int64_t t0 = (bid - b * P.tiles_per_clip) * G;
    const int Gt = (int)((P.T - t0) < G ? (P.T - t0) : G);
    const float *yb = P.y + b * P.L;

    // ---- load: virtual pad + frame + window, packed as complex -----------------
    for (int item = tid; item < G * nc; item += nthreads) {
        const int g = item / nc;
        const int c = item - g * nc;
        ap_float2 z = ap_mk(0.0f, 0.0f);
        if (g < Gt) {
            const int64_t base = (t0 + g) * (int64_t)P.hop - P.pad;
            if (pl.even) {
                const int s = 2 * c;
                z.x = P.window[s] * ap_load_padded(yb, P.L, base + s, P.pad_mode);
                z.y = P.window[s + 1] * ap_load_padded(yb, P.L, base + s + 1, P.pad_mode);
            } else {
                z.x = P.window[c] * ap_load_padded(yb, P.L, base + c, P.pad_mode);
            }
        }
        bufA[g * fstride + c] = z;
    }
    AP_LDS_BARRIER();

    ap_float2 *Z = ap_fft_tile(bufA, bufB, pl, P.tw, G, fstride, tid, nthreads);
    const int F = P.n_bins;

    if (EPI == 0) {
        // transposed store: frames of the tile are the fastest-varying lanes
        for (int item = tid; item < F * G; item += nthreads) {
            const int k = item / G;
            const int g = item - k * G;
            if (g < Gt) {
                const ap_float2 *Zg = Z + g * fstride;
                const ap_float2 X = pl.even ? ap_rfft_split(Zg, nc, k, P.tw) : Zg[k];
                P.out_c[(b * F + k) * P.T + t0 + g] = X;
            }
        }
    } else {
        float *Pw = reinterpret_cast<float *>(Z == bufA ? bufB : bufA);
        const int pstride = 2 * fstride;
        for (int item = tid; item < F * G; item += nthreads) {
            const int k = item / G;
            const int g = item - k * G;
            const ap_float2 *Zg = Z + g * fstride;
            const ap_float2 X = pl.even ? ap_rfft_split(Zg, nc, k, P.tw) : Zg[k];
            Pw[g * pstride + k] = ap_pow_mag(X.x, X.y, P.power);
        }
        AP_LDS_BARRIER();
        // banded contraction: zeros outside [lo, lo+len) contribute exactly 0
        for (int item = tid; item < P.n_mels * G; item += nthreads) {
            const int m = item / G;
            const int g = item - m * G;
            if (g < Gt) {
                const int lo = P.band_lo ? P.band_lo[m] : 0;
                const int len = P.band_len ? P.band_len[m] : F;
                const float *w = P.fb + (int64_t)m * F + lo;
                const float *pp = Pw + g * pstride + lo;
                float acc = 0.0f;
                for (int i = 0; i < len; ++i) acc = fmaf(w[i], pp[i], acc);
                P.out_mel[(b * P.n_mels + m) * P.T + t0 + g] = acc;
            }
        }
    }
    (void)n;
}

// irfft of each frame: S (B,F,T) -> frames (B,T,n).  Inverse via the forward
// engine on conjugated data.
AP_KERNEL void __launch_bounds__(AP_BLOCK) ap_irfft_generic_kernel(ApIrfftParams P) {
    const ApFftPlan &pl = P.plan;
    const int G = P.tile.G, fstride = P.tile.fstride, nc = pl.nc, n = pl.n;
    ap_float2 *bufA = reinterpret_cast<ap_float2 *>(ap_smem);
    ap_float2 *bufB = bufA + (size_t)G * fstride;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int64_t bid = blockIdx.x;
    const int64_t b = bid / P.tiles_per_clip;
    const int64_t t0 = (bid - b * P.tiles_per_clip) * G;
    const int Gt = (int)((P.T - t0) < G ? (P.T - t0) : G);
    const int F = P.n_bins;
    const ap_float2 *Sb = P.S + b * F * P.T;

    for (int item = tid; item < nc * G; item += nthreads) {
        const int k = item / G;
        const int g = item - k * G;
        ap_float2 zc = ap_mk(0.0f, 0.0f);
        if (g < Gt) {
            const int64_t col = t0 + g;
            if (pl.even) {
                ap_float2 xk = Sb[(int64_t)k * P.T + col];
                ap_float2 xm = Sb[(int64_t)(nc - k) * P.T + col];
                if (k == 0) { xk.y = 0.0f; xm.y = 0.0f; }   // DC / Nyquist imaginary parts ignored
                const float ax = xk.x + xm.x, ay = xk.y - xm.y;
                const float dx = xk.x - xm.x, dy = xk.y + xm.y;
                const ap_float2 w = P.tw[k];               // W_n^{-k} = (c, +s)
                const float ox = w.x * dx - w.y * dy, oy = w.x * dy + w.y * dx;
                zc = ap_mk(ax - oy, -(ay + ox));           // conj(E + iO) (x2, folded into 1/n)
            } else {
                const int half = (n - 1) / 2;
                if (k <= half) {
                    ap_float2 x = Sb[(int64_t)k * P.T + col];
                    if (k == 0) x.y = 0.0f;
                    zc = ap_mk(x.x, -x.y);
                } else {
                    zc = Sb[(int64_t)(n - k) * P.T + col];  // conj(conj(X[n-k]))
                }
            }
        }
        bufA[g * fstride + k] = zc;
    }
    AP_LDS_BARRIER();

    ap_float2 *Y = ap_fft_tile(bufA, bufB, pl, P.tw, G, fstride, tid, nthreads);
    const float scale = 1.0f / (float)n;
    for (int item = tid; item < G * nc; item += nthreads) {
        const int g = item / nc;
        const int c = item - g * nc;
        if (g < Gt) {
            const ap_float2 v = Y[g * fstride + c];
            float *dst = P.frames + ((b * P.T + t0 + g) * (int64_t)n);
            if (pl.even) {
                dst[2 * c] = v.x * scale;
                dst[2 * c + 1] = -v.y * scale;
            } else {
                dst[c] = v.x * scale;
            }
        }
    }
}

// overlap_add.metal:16-55 with an output offset (folds istft's centre trim).
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_overlap_add_kernel(const float *frames, const float *window, int64_t T, int n_fft, int hop,
                      int64_t out_offset, int64_t out_len, int64_t blocks_per_row, float *out) {
    const int64_t bid = blockIdx.x;
    const int64_t b = bid / blocks_per_row;
    const int64_t i = (bid - b * blocks_per_row) * blockDim.x + threadIdx.x;
    if (i >= out_len) return;
    const int64_t p = i + out_offset;
    int64_t first = (p - n_fft + 1 + hop - 1) / hop;   // ceil((p-N+1)/hop) for the non-negative case
    if (p - n_fft + 1 <= 0) first = 0;
    int64_t last = p / hop;
    if (last >= T) last = T - 1;
    float sum = 0.0f, wss = 0.0f;
    const float *fb = frames + b * T * n_fft;
    for (int64_t f = first; f <= last; ++f) {
        const int s = (int)(p - f * hop);
        const float w = window[s];
        sum += w * fb[f * n_fft + s];
        wss += w * w;
    }
    out[b * out_len + i] = sum / fmaxf(wss, 1e-8f);
}

AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_pad_kernel(const float *x, int64_t B, int64_t L, int64_t pad, int mode, float *out) {
    const int64_t Lo = L + 2 * pad;
    const int64_t total = B * Lo;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / Lo;
        const int64_t i = e - b * Lo;
        out[e] = ap_load_padded(x + b * L, L, i - pad, mode);
    }
}

AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_frame_kernel(const float *x, int64_t B, int64_t L, int64_t T, int frame_length, int hop,
                float *out) {
    const int64_t total = B * T * frame_length;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t s = e % frame_length;
        const int64_t bt = e / frame_length;
        const int64_t t = bt % T;
        const int64_t b = bt / T;
        out[e] = x[b * L + t * hop + s];
    }
}

// mode 0: |S| (mx.abs, stft.py:362)   mode 1: atan2(im, re) (stft.py:379)
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_complex_unary_kernel(const ap_float2 *S, int64_t n, int mode, float *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const ap_float2 v = S[e];
        out[e] = mode == 0 ? sqrtf(v.x * v.x + v.y * v.y) : atan2f(v.y, v.x);
    }
}

// the same from a spectrum whose rows are Ts complex values apart (ap_stft_rows_f32) into a DENSE (rows, T) output
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_complex_unary_rows_kernel(const ap_float2 *S, int64_t rows, int T, int Ts, int mode, float *out) {
    const int64_t n = rows * T;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int64_t row = e / T;
        const ap_float2 v = S[row * Ts + (e - row * T)];
        out[e] = mode == 0 ? sqrtf(v.x * v.x + v.y * v.y) : atan2f(v.y, v.x);
    }
}

// scipy.signal.resample_poly (upfirdn, zero padding) — reference resample.py:279-281.
// One thread per output sample; float32 accumulation in increasing input index.
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_resample_poly_kernel(const float *x, int64_t L, int up, int down, const float *taps, int n_taps,
                        int n_pre_remove, int64_t n_out, int64_t blocks_per_row, float *out) {
#ifndef AP_HOST_EMU
#pragma clang fp contract(off)   // SciPy's C loop rounds the product, then the sum: no FMA
#endif
    const int64_t bid = blockIdx.x;
    const int64_t b = bid / blocks_per_row;
    const int64_t o = (bid - b * blocks_per_row) * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int64_t t = (o + n_pre_remove) * (int64_t)down;
    int64_t i_hi = t / up;
    if (i_hi > L - 1) i_hi = L - 1;
    int64_t num = t - (n_taps - 1);
    int64_t i_lo = num <= 0 ? 0 : (num + up - 1) / up;
    const float *xb = x + b * L;
    float acc = 0.0f;
    for (int64_t i = i_lo; i <= i_hi; ++i) acc = acc + taps[t - (int64_t)up * i] * xb[i];
    out[b * n_out + o] = acc;
}

// The same for up > 1 (44.1 kHz <-> 48 kHz: 160 / 147) with the filter and the input span in LDS: a
// workgroup owns 256 R consecutive outputs, stages the input samples they touch once (zeros outside the
// clip = SciPy's zero padding) and the polyphase table hp[ph][k] = taps[ph + k up] (zero past the filter;
// row stride odd); output o with t = (o + n_pre_remove) down, ph = t mod up, i = t div up is
// sum_k hp[ph][k] x[i - k], accumulated from the largest k down = increasing input index, the product
// rounded before the sum: the order and roundings of SciPy's loop, so still bit-exact.
#define AP_RSPL_R 4
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_resample_poly_lds_kernel(const float *x, int64_t L, int up, int down, const float *taps, int n_taps,
                            int n_pre_remove, int64_t n_out, int64_t blocks_per_row, int K, int KS, int span,
                            float *out) {
#ifndef AP_HOST_EMU
#pragma clang fp contract(off)   // SciPy's C loop rounds the product, then the sum: no FMA
#endif
    float *hp = reinterpret_cast<float *>(ap_smem);          // [up][KS]
    float *xs = hp + (size_t)up * KS;                        // [span]
    const int tid = threadIdx.x;
    const int64_t bid = blockIdx.x;
    const int64_t b = bid / blocks_per_row;
    const int64_t o0 = (bid - b * blocks_per_row) * (int64_t)(AP_BLOCK * AP_RSPL_R);
    const float *xb = x + b * L;
    // first input sample any output of the block touches: i(o0) - (K - 1)
    const int64_t lo = ((o0 + n_pre_remove) * (int64_t)down) / up - (K - 1);
    for (int i = tid; i < up * KS; i += AP_BLOCK) {
        const int ph = i / KS, k = i - ph * KS;
        const int64_t j = ph + (int64_t)k * up;
        hp[i] = (k < K && j < n_taps) ? taps[j] : 0.0f;
    }
    for (int i = tid; i < span; i += AP_BLOCK) {
        const int64_t n = lo + i;
        xs[i] = (n >= 0 && n < L) ? xb[n] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < AP_RSPL_R; ++r) {
        const int64_t o = o0 + tid + (int64_t)AP_BLOCK * r;
        if (o >= n_out) break;
        const int64_t t = (o + n_pre_remove) * (int64_t)down;
        const int64_t iq = t / up;
        const int ph = (int)(t - iq * up);
        const float *h = hp + ph * KS;
        const float *xr = xs + (int)(iq - lo);               // x[i]; x[i - k] = xr[-k]
        float acc = 0.0f;
        for (int k = K - 1; k >= 0; --k) acc = acc + h[k] * xr[-k];
        out[b * n_out + o] = acc;
    }
}

// Decimating case (up == 1, e.g. 48 kHz -> 16 kHz) of the same filter, LDS-tiled and register
// blocked: a workgroup stages the contiguous input span of 256*R outputs once (coalesced,
// zeros outside the clip = SciPy's zero padding) and every thread slides over the span
// accumulating R = 4 consecutive outputs.  Per 4 input samples: one 16-byte LDS read of the
// span, four 16-byte broadcast reads of the tap table hs4[d] = the taps outputs 0..3 apply to
// sample d (zero outside the filter), and 8 + 8 packed multiplies / adds (two outputs per
// instruction).  Same products, same increasing-input-index order and no FMA contraction:
// still bit-exact against SciPy.
#define AP_RSP_R 4
struct __attribute__((aligned(16))) ap_rsp_f4 { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) ap_rsp_f4u { float x, y, z, w; };   // 4-byte aligned 16-byte load
// two outputs per instruction; the multiply and the add are spelled inside the kernel so that
// its `fp contract(off)` covers them (hipcc's default would fuse them into one rounding)
#ifdef AP_PACKED_COMPLEX
#define AP_RSP_MAC(acc, hx, hy, xv) acc = acc + ap_mk(hx, hy) * ap_mk(xv, xv)
#else
#define AP_RSP_MAC(acc, hx, hy, xv) do { acc.x = acc.x + (hx) * (xv); acc.y = acc.y + (hy) * (xv); } while (0)
#endif

// Q = quad-groups of outputs per thread (outputs o0 + 1024 q + 4 tid + r): one tap read serves Q
// groups, which moves the loop from LDS-issue bound to VALU bound
// STEPS > 0: the window length is a compile-time constant (72 = the 61-tap 3:1 filter of 48 kHz -> 16 kHz): the
// loop over the window is unrolled and its LDS reads run ahead of the arithmetic
template <int Q, int STEPS = 0>
__global__ void __launch_bounds__(AP_BLOCK)
ap_resample_decim_kernel(const float *x, int64_t L, int down, const float *taps, int n_taps,
                         int n_pre_remove, int64_t n_out, int64_t blocks_per_row, float *out) {
#ifndef AP_HOST_EMU
#pragma clang fp contract(off)
#endif
    const int tid = threadIdx.x;
    const int R = AP_RSP_R;
    const int margin = down * (R - 1);
    const int steps = STEPS > 0 ? STEPS : ((n_taps + margin + 3) & ~3);   // d = 0 .. n_taps-1+margin, rounded up to 4
    ap_rsp_f4 *hs4 = reinterpret_cast<ap_rsp_f4 *>(ap_smem);
    float *xs = reinterpret_cast<float *>(hs4 + steps);
    const int64_t bid = blockIdx.x;
    const int64_t b = bid / blocks_per_row;
    const int64_t o0 = (bid - b * blocks_per_row) * (AP_BLOCK * R * Q);
    const int span = AP_BLOCK * R * Q * down + steps;       // input samples this workgroup touches (+ tail)
    const int64_t s0 = (o0 + n_pre_remove) * (int64_t)down - (n_taps - 1);   // first of them (may be < 0)
    const float *xb = x + b * L;
    for (int i = tid; i < steps * R; i += AP_BLOCK) {
        // output r sees sample d through tap (n_taps-1) - d + down*r (zero outside the filter)
        const int d = i >> 2, r = i & 3;
        const int j = (n_taps - 1) - d + down * r;
        reinterpret_cast<float *>(hs4)[i] = (j >= 0 && j < n_taps) ? taps[j] : 0.0f;
    }
    // stage the span 4 samples per thread and step (span is a multiple of 4; LDS side 16-byte
    // aligned, global side only 4-byte aligned)
    for (int i = 4 * tid; i < span; i += 4 * AP_BLOCK) {
        const int64_t g = s0 + i;
        ap_rsp_f4 v;
        if (g >= 0 && g + 3 < L) {
            const ap_rsp_f4u u = *reinterpret_cast<const ap_rsp_f4u *>(xb + g);
            v.x = u.x; v.y = u.y; v.z = u.z; v.w = u.w;
        } else {
            v.x = (g >= 0 && g < L) ? xb[g] : 0.0f;
            v.y = (g + 1 >= 0 && g + 1 < L) ? xb[g + 1] : 0.0f;
            v.z = (g + 2 >= 0 && g + 2 < L) ? xb[g + 2] : 0.0f;
            v.w = (g + 3 >= 0 && g + 3 < L) ? xb[g + 3] : 0.0f;
        }
        *reinterpret_cast<ap_rsp_f4 *>(xs + i) = v;
    }
    AP_LDS_BARRIER();
    // group q of this thread starts its window at xs[(1024 q + 4 tid) * down] (16-byte aligned)
    ap_float2 acc01[Q], acc23[Q];
    const ap_rsp_f4 *xq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        acc01[q] = ap_mk(0.0f, 0.0f);
        acc23[q] = ap_mk(0.0f, 0.0f);
        xq[q] = reinterpret_cast<const ap_rsp_f4 *>(xs + (AP_BLOCK * R * q + R * tid) * down);
    }
#pragma clang loop unroll_count(STEPS > 0 ? STEPS / 4 : 1)
    for (int d = 0; d < steps; d += 4) {
        const ap_rsp_f4 h0 = hs4[d], h1 = hs4[d + 1], h2 = hs4[d + 2], h3 = hs4[d + 3];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const ap_rsp_f4 xv = xq[q][d >> 2];
            AP_RSP_MAC(acc01[q], h0.x, h0.y, xv.x);
            AP_RSP_MAC(acc23[q], h0.z, h0.w, xv.x);
            AP_RSP_MAC(acc01[q], h1.x, h1.y, xv.y);
            AP_RSP_MAC(acc23[q], h1.z, h1.w, xv.y);
            AP_RSP_MAC(acc01[q], h2.x, h2.y, xv.z);
            AP_RSP_MAC(acc23[q], h2.z, h2.w, xv.z);
            AP_RSP_MAC(acc01[q], h3.x, h3.y, xv.w);
            AP_RSP_MAC(acc23[q], h3.z, h3.w, xv.w);
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t o = o0 + (int64_t)(AP_BLOCK * R) * q + (int64_t)R * tid;
        float *dst = out + b * n_out + o;
        if (o + 3 < n_out && ((b * n_out + o) & 3) == 0) {
            ap_rsp_f4 v;
            v.x = acc01[q].x; v.y = acc01[q].y; v.z = acc23[q].x; v.w = acc23[q].y;
            *reinterpret_cast<ap_rsp_f4 *>(dst) = v;
        } else {
            if (o < n_out) dst[0] = acc01[q].x;
            if (o + 1 < n_out) dst[1] = acc01[q].y;
            if (o + 2 < n_out) dst[2] = acc23[q].x;
            if (o + 3 < n_out) dst[3] = acc23[q].y;
        }
    }
}

// The same with TWO outputs per thread and group: their shared window is n_taps + down samples instead of n_taps +
// 3 down, so fewer of the window's products are the zero-tap ones (61 taps, down 3: 64 positions per output instead
// of 72).  The kernel runs at the energy ceiling of its packed multiplies and adds, so the instruction
// count is its time.  Same products, same order, no FMA contraction: bit-exact against SciPy like the kernel above.
template <int Q>
__global__ void __launch_bounds__(AP_BLOCK)
ap_resample_decim2_kernel(const float *x, int64_t L, int down, const float *taps, int n_taps,
                          int n_pre_remove, int64_t n_out, int64_t blocks_per_row, float *out) {
#ifndef AP_HOST_EMU
#pragma clang fp contract(off)
#endif
    const int tid = threadIdx.x;
    const int R = 2;
    const int steps = (n_taps + down * (R - 1) + 3) & ~3;    // d = 0 .. n_taps-1+down, rounded up to 4
    ap_rsp_f4 *hs2 = reinterpret_cast<ap_rsp_f4 *>(ap_smem);  // pairs of samples: (tap of output 0, of output 1) x 2
    float *xs = reinterpret_cast<float *>(ap_smem) + steps * R;
    const int64_t bid = blockIdx.x;
    const int64_t b = bid / blocks_per_row;
    const int64_t o0 = (bid - b * blocks_per_row) * (AP_BLOCK * R * Q);
    const int span = AP_BLOCK * R * Q * down + steps;
    const int64_t s0 = (o0 + n_pre_remove) * (int64_t)down - (n_taps - 1);
    const float *xb = x + b * L;
    for (int i = tid; i < steps * R; i += AP_BLOCK) {
        const int d = i >> 1, r = i & 1;
        const int j = (n_taps - 1) - d + down * r;            // output r sees sample d through tap j
        reinterpret_cast<float *>(hs2)[i] = (j >= 0 && j < n_taps) ? taps[j] : 0.0f;
    }
    for (int i = 4 * tid; i < span; i += 4 * AP_BLOCK) {
        const int64_t g = s0 + i;
        ap_rsp_f4 v;
        if (g >= 0 && g + 3 < L) {
            const ap_rsp_f4u u = *reinterpret_cast<const ap_rsp_f4u *>(xb + g);
            v.x = u.x; v.y = u.y; v.z = u.z; v.w = u.w;
        } else {
            v.x = (g >= 0 && g < L) ? xb[g] : 0.0f;
            v.y = (g + 1 >= 0 && g + 1 < L) ? xb[g + 1] : 0.0f;
            v.z = (g + 2 >= 0 && g + 2 < L) ? xb[g + 2] : 0.0f;
            v.w = (g + 3 >= 0 && g + 3 < L) ? xb[g + 3] : 0.0f;
        }
        *reinterpret_cast<ap_rsp_f4 *>(xs + i) = v;
    }
    AP_LDS_BARRIER();
    // group q of this thread starts its window at xs[(512 q + 2 tid) * down] (8-byte aligned)
    ap_float2 acc[Q];
    const ap_float2 *xq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        acc[q] = ap_mk(0.0f, 0.0f);
        xq[q] = reinterpret_cast<const ap_float2 *>(xs + (AP_BLOCK * R * q + R * tid) * down);
    }
    for (int d = 0; d < steps; d += 4) {
        const ap_rsp_f4 h01 = hs2[d >> 1], h23 = hs2[(d >> 1) + 1];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const ap_float2 xa = xq[q][d >> 1], xb2 = xq[q][(d >> 1) + 1];
            AP_RSP_MAC(acc[q], h01.x, h01.y, xa.x);
            AP_RSP_MAC(acc[q], h01.z, h01.w, xa.y);
            AP_RSP_MAC(acc[q], h23.x, h23.y, xb2.x);
            AP_RSP_MAC(acc[q], h23.z, h23.w, xb2.y);
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t o = o0 + (int64_t)(AP_BLOCK * R) * q + (int64_t)R * tid;
        float *dst = out + b * n_out + o;
        if (o + 1 < n_out && ((b * n_out + o) & 1) == 0) {
            *reinterpret_cast<ap_float2 *>(dst) = acc[q];
        } else {
            if (o < n_out) dst[0] = acc[q].x;
            if (o + 1 < n_out) dst[1] = acc[q].y;
        }
    }
}

// reference resample.py:183-195: float64 positions and interpolation, float32 result
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_resample_linear_kernel(const float *x, int64_t B, int64_t L, int64_t n_out, double scale, float *out) {
    const int64_t total = B * n_out;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double step = n_out > 1 ? (double)(L - 1) / (double)(n_out - 1) : 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / n_out;
        const int64_t j = e - b * n_out;
        double pos = (double)j * step;
        if (j == n_out - 1 && n_out > 1) pos = (double)(L - 1);
        int64_t lo = (int64_t)floor(pos);
        int64_t hi = lo + 1 < L ? lo + 1 : L - 1;
        const double frac = pos - (double)lo;
        const double v = (1.0 - frac) * (double)x[b * L + lo] + frac * (double)x[b * L + hi];
        out[e] = (float)(v * scale);
    }
}
