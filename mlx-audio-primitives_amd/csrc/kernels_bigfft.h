// Large-N complex FFT for scipy.signal.resample (reference resample.py:97,123: one whole-clip
// FFT of arbitrary length on the host).  Four-step (Bailey) transform, N = N1*N2 with both
// legs LDS-resident: leg 1 = N2 strided transforms of length N1 + the W_N^(n2*k1) twiddle,
// leg 2 = N1 contiguous transforms of length N2 with a strided (transposing) store.  Both
// legs are this one kernel; a workgroup takes G adjacent frames so the strided side of each
// leg is still read / written in G-element contiguous pieces.
#pragma once
#include "fft_lds.h"

#ifdef AP_HOST_EMU
#include <cmath>
AP_DEV void ap_sincos_2pi(double frac, float *s, float *c) {
    const long double a = 6.283185307179586476925286766559L * (long double)frac;
    *s = (float)sinl(a);
    *c = (float)cosl(a);
}
#else
// float32 sincospi of a double-precision turn fraction: the argument is exact to 6e-8 turns
// (3.7e-7 rad), the evaluation to an ulp - against a float64 sincospi this costs < 1e-6 relative on
// the twiddle and saves ~100 instructions per element of the twiddled leg
AP_DEV void ap_sincos_2pi(double frac, float *s, float *c) { sincospif((float)(2.0 * frac), s, c); }
#endif

__global__ void __launch_bounds__(AP_BLOCK) ap_cfft_strided_kernel(ApCfftParams P) {
    const ApFftPlan &pl = P.plan;
    const int G = P.tile.G, fstride = P.tile.fstride, n = pl.nc;
    ap_float2 *bufA = reinterpret_cast<ap_float2 *>(ap_smem);
    ap_float2 *bufB = bufA + (size_t)G * fstride;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int64_t bid = blockIdx.x;
    const int64_t sig = bid / P.tiles_per_signal;
    const int64_t f0 = (bid - sig * P.tiles_per_signal) * G;
    const int Gt = (int)((P.n_frames - f0) < G ? (P.n_frames - f0) : G);
    const float *in_r = reinterpret_cast<const float *>(P.in) + sig * P.in_batch;
    const ap_float2 *in_c = reinterpret_cast<const ap_float2 *>(P.in) + sig * P.in_batch;
    const bool frames_fast = P.in_fs == 1;       // adjacent frames contiguous -> lanes run over frames

    for (int item = tid; item < G * n; item += nthreads) {
        int g, i;
        if (frames_fast) { i = item / G; g = item - i * G; } else { g = item / n; i = item - g * n; }
        ap_float2 z = ap_mk(0.0f, 0.0f);
        if (g < Gt) {
            const int64_t a = (f0 + g) * P.in_fs + (int64_t)i * P.in_is;
            if (P.real_in) z.x = in_r[a]; else z = in_c[a];
            if (P.conj_io) z.y = -z.y;
        }
        bufA[g * fstride + i] = z;
    }
    AP_LDS_BARRIER();
    ap_float2 *Z = ap_fft_tile(bufA, bufB, pl, P.tw, G, fstride, tid, nthreads);

    float *out_r = reinterpret_cast<float *>(P.out) + sig * P.out_batch;
    ap_float2 *out_c = reinterpret_cast<ap_float2 *>(P.out) + sig * P.out_batch;
    const bool oframes_fast = P.out_fs == 1;
    for (int item = tid; item < G * n; item += nthreads) {
        int g, k;
        if (oframes_fast) { k = item / G; g = item - k * G; } else { g = item / n; k = item - g * n; }
        if (g < Gt) {
            ap_float2 v = Z[g * fstride + k];
            if (P.tw_N > 0) {
                // n2 k1 < N2 N1 = N: no reduction needed, and the product fits 32 bits (both <= 4096)
                const unsigned m = (unsigned)(f0 + g) * (unsigned)k;
                float s, c;
                ap_sincos_2pi((double)m / (double)P.tw_N, &s, &c);
                v = ap_mul_fw(v, ap_mk(c, s));              // * exp(-2 pi i m / N)
            }
            if (P.conj_io) v.y = -v.y;
            v.x *= P.scale;
            v.y *= P.scale;
            const int64_t a = (f0 + g) * P.out_fs + (int64_t)k * P.out_is;
            if (P.real_out) out_r[a] = v.x; else out_c[a] = v;
        }
    }
}

// scipy.signal.resample's spectrum surgery for real input (SciPy 1.15 _signaltools.resample):
// keep the N//2+1 lowest bins (N = min(num, Nx)), scale the shared Nyquist bin by 2 (down) or
// 0.5 (up), zero the rest, and lay the result out as the FULL Hermitian spectrum of length
// num for the inverse complex transform; the imaginary part of bin 0 (and num/2) is dropped
// as irfft does.
__global__ void __launch_bounds__(AP_BLOCK)
ap_resample_spectrum_kernel(const ap_float2 *X, int64_t Nx, ap_float2 *Y, int64_t num, int64_t B) {
    const int64_t total = B * num;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t N = num < Nx ? num : Nx;
    const int64_t nyq = N / 2 + 1;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / num, k = e - b * num;
        const int64_t kk = k <= num / 2 ? k : num - k;         // Hermitian partner in [0, num/2]
        ap_float2 v = ap_mk(0.0f, 0.0f);
        if (kk < nyq) {
            v = X[b * Nx + kk];
            if (N % 2 == 0 && kk == N / 2) {
                if (num < Nx) { v.x *= 2.0f; v.y *= 2.0f; }
                else if (Nx < num) { v.x *= 0.5f; v.y *= 0.5f; }
            }
            if (kk == 0 || (num % 2 == 0 && kk == num / 2)) v.y = 0.0f;
            if (k != kk) v.y = -v.y;
        }
        Y[e] = v;
    }
}

// ---- chirp-z (Bluestein) wrapping of the four-step transform ------------------------------------
// A length-N DFT for N with a prime factor the LDS legs cannot hold, as a circular convolution of a
// supported length M >= 2N - 1:  n k = (n^2 + k^2 - (k - n)^2) / 2, so with c[n] = exp(-i pi n^2 / N)
//   X[k] = c[k] * sum_n (x[n] c[n]) * conj(c)[k - n]
// pre:  a[n] = x[n] c[n] (n < N), 0 up to M;  then A = FFT_M(a), A *= FFT_M(conj c, wrapped) (a host-built
// table), a' = IFFT_M(A);  post: X[k] = c[k] a'[k].  The inverse transform conjugates all three factors.
__global__ void __launch_bounds__(AP_BLOCK)
ap_chirp_pre_kernel(const void *in, int real_in, int64_t N, const ap_float2 *chirp, int conj, ap_float2 *out,
                    int64_t M, int64_t B) {
    const int64_t total = B * M;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / M, n = e - b * M;
        ap_float2 v = ap_mk(0.0f, 0.0f);
        if (n < N) {
            const ap_float2 x = real_in ? ap_mk(reinterpret_cast<const float *>(in)[b * N + n], 0.0f)
                                        : reinterpret_cast<const ap_float2 *>(in)[b * N + n];
            ap_float2 c = chirp[n];
            if (conj) c.y = -c.y;
            v = ap_mk(x.x * c.x - x.y * c.y, x.x * c.y + x.y * c.x);
        }
        out[e] = v;
    }
}

__global__ void __launch_bounds__(AP_BLOCK)
ap_chirp_spec_kernel(ap_float2 *buf, const ap_float2 *spec, int conj, int64_t M, int64_t B) {
    const int64_t total = B * M;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const ap_float2 x = buf[e];
        ap_float2 c = spec[e % M];
        if (conj) c.y = -c.y;
        buf[e] = ap_mk(x.x * c.x - x.y * c.y, x.x * c.y + x.y * c.x);
    }
}

__global__ void __launch_bounds__(AP_BLOCK)
ap_chirp_post_kernel(const ap_float2 *buf, int64_t M, const ap_float2 *chirp, int conj, int64_t N, float scale,
                     int real_out, void *out, int64_t B) {
    const int64_t total = B * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / N, k = e - b * N;
        const ap_float2 x = buf[b * M + k];
        ap_float2 c = chirp[k];
        if (conj) c.y = -c.y;
        const ap_float2 v = ap_mk((x.x * c.x - x.y * c.y) * scale, (x.x * c.y + x.y * c.x) * scale);
        if (real_out) reinterpret_cast<float *>(out)[e] = v.x;
        else reinterpret_cast<ap_float2 *>(out)[e] = v;
    }
}
