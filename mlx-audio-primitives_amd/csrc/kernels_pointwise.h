// Element-wise / small-contraction kernels of the hot path's tails: Griffin-Lim phase
// projection (griffinlim.py:123,168-178), dB conversions with the reference's GLOBAL
// max clip (convert.py:42-60), the DCT-II contraction of mfcc (mfcc.py:135, dct.cpp:148).
// All are HBM-bound streaming kernels: grid-stride, one element per lane per step.
#pragma once
#include "fft_lds.h"

// mode 0: out = S * exp(i * angles)                        (griffinlim.py:123, init)
// mode 1: R' = S * exp(i * atan2(R.im, R.re));  rebuilt = R' + m (R' - tprev); tprev = R'
//         R is (B,F,TR); frames t >= TR are treated as zero (griffinlim.py:156-165 crop/pad)
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_gl_project_kernel(int mode, const float *S, const float *angles, const ap_float2 *R, int64_t TR,
                     int64_t BF, int64_t T, float momentum, ap_float2 *tprev, ap_float2 *rebuilt) {
    const int64_t total = BF * T;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        float ang;
        if (mode == 0) {
            ang = angles[e];
        } else {
            const int64_t row = e / T, t = e - row * T;
            ap_float2 r = ap_mk(0.0f, 0.0f);
            if (t < TR) r = R[row * TR + t];
            ang = atan2f(r.y, r.x);
        }
        const float s = S[e];
        const ap_float2 rn = ap_mk(s * cosf(ang), s * sinf(ang));
        if (mode == 0) {
            rebuilt[e] = rn;
            if (tprev) tprev[e] = rn;
        } else if (momentum > 0.0f) {
            const ap_float2 tp = tprev[e];
            rebuilt[e] = ap_mk(rn.x + momentum * (rn.x - tp.x), rn.y + momentum * (rn.y - tp.y));
            tprev[e] = rn;
        } else {
            rebuilt[e] = rn;
        }
    }
}

// mode 1 without the tprev array: tprev = S exp(i angle(R_prev)) is a function of the previous raw
// STFT, so the loop keeps two raw buffers (ping-pong) and this pass reads R_cur, R_prev and S and
// writes only `rebuilt` (28 instead of 36 bytes per element).  exp(i angle(R)) = R / |R| (1 for
// R = 0, like atan2(0, 0) = 0).  Needs TR == T.
// 1 / |r| on the hardware's reciprocal square root (v_rsq_f32, 1 ulp), branch-free; tiny values are scaled up first
// (|r|^2 would be subnormal below ~1e-19).  The fused Griffin-Lim kernels (kernels_stft16.h) use the same function,
// so both routes give the same bits.
AP_DEV float ap_rnorm(ap_float2 &r) {
    float n2 = r.x * r.x + r.y * r.y;
    const bool tiny = n2 < 1.0e-30f;
    r.x = tiny ? r.x * 1.8446744e19f : r.x;            // 2^64
    r.y = tiny ? r.y * 1.8446744e19f : r.y;
    n2 = tiny ? r.x * r.x + r.y * r.y : n2;
#ifdef AP_HOST_EMU
    return 1.0f / sqrtf(n2);
#else
    return __builtin_amdgcn_rsqf(n2);
#endif
}
AP_DEV ap_float2 ap_unit_phase(ap_float2 r) {
    const float inv = ap_rnorm(r);
    const bool zero = !(r.x != 0.0f || r.y != 0.0f);    // exp(i atan2(0, 0)) = 1 (NaN inputs propagate)
    return ap_mk(zero ? 1.0f : r.x * inv, zero ? 0.0f : r.y * inv);
}

AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_gl_project2_kernel(const float *S, const ap_float2 *Rcur, const ap_float2 *Rprev, int64_t total,
                      float momentum, ap_float2 *rebuilt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const float s = S[e];
        const ap_float2 u = ap_unit_phase(Rcur[e]);
        const ap_float2 rn = ap_mk(s * u.x, s * u.y);
        if (momentum > 0.0f) {
            const ap_float2 v = ap_unit_phase(Rprev[e]);
            const ap_float2 tp = ap_mk(s * v.x, s * v.y);
            rebuilt[e] = ap_mk(rn.x + momentum * (rn.x - tp.x), rn.y + momentum * (rn.y - tp.y));
        } else {
            rebuilt[e] = rn;
        }
    }
}

// The same two passes on workspaces whose rows are Ts complex values apart (Ts even, 16-byte aligned rows:
// the line-padded layout of ap_stft_rows_f32 / ap_istft_rows_f32); S and `angles` stay dense (rows of T).
// One thread per PAIR of frames: 16-byte accesses on the workspaces.  The padding columns are never touched.
//   angles != NULL : rebuilt = tprev = S exp(i angles)                       (griffinlim.py:123, init)
//   angles == NULL : rebuilt = R' + m (R' - S unit(Rprev)),  R' = S unit(Rcur)   (griffinlim.py:168-178)
template <class IDX>
__global__ void __launch_bounds__(AP_BLOCK)
ap_gl_rows_kernel(const float *S, const float *angles, const ap_float2 *Rcur, const ap_float2 *Rprev, int64_t rows,
                  int T, int Ts, float momentum, ap_float2 *tprev, ap_float2 *rebuilt) {
    const IDX half = (IDX)(Ts >> 1);
    const IDX total = (IDX)rows * half;
    const IDX stride = (IDX)gridDim.x * blockDim.x;
    for (IDX e = (IDX)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const IDX row = e / half;
        const int t = (int)(e - row * half) * 2;
        if (t >= T) continue;
        const bool two = t + 1 < T;
        const int64_t w = (int64_t)row * Ts + t, d = (int64_t)row * T + t;
        const float s0 = S[d], s1 = two ? S[d + 1] : 0.0f;
        ap_float2 o0, o1;
        if (angles) {
            const float a0 = angles[d], a1 = two ? angles[d + 1] : 0.0f;
            o0 = ap_mk(s0 * cosf(a0), s0 * sinf(a0));
            o1 = ap_mk(s1 * cosf(a1), s1 * sinf(a1));
        } else {
            const ap_float4 c = *reinterpret_cast<const ap_float4 *>(Rcur + w);
            const ap_float2 u0 = ap_unit_phase(ap_mk(c.x, c.y)), u1 = ap_unit_phase(ap_mk(c.z, c.w));
            o0 = ap_mk(s0 * u0.x, s0 * u0.y);
            o1 = ap_mk(s1 * u1.x, s1 * u1.y);
            if (momentum > 0.0f) {
                const ap_float4 p = *reinterpret_cast<const ap_float4 *>(Rprev + w);
                const ap_float2 v0 = ap_unit_phase(ap_mk(p.x, p.y)), v1 = ap_unit_phase(ap_mk(p.z, p.w));
                o0 = ap_mk(o0.x + momentum * (o0.x - s0 * v0.x), o0.y + momentum * (o0.y - s0 * v0.y));
                o1 = ap_mk(o1.x + momentum * (o1.x - s1 * v1.x), o1.y + momentum * (o1.y - s1 * v1.y));
            }
        }
        if (two) {
            ap_float4 o; o.x = o0.x; o.y = o0.y; o.z = o1.x; o.w = o1.y;
            *reinterpret_cast<ap_float4 *>(rebuilt + w) = o;
            if (tprev) *reinterpret_cast<ap_float4 *>(tprev + w) = o;
        } else {
            rebuilt[w] = o0;
            if (tprev) tprev[w] = o0;
        }
    }
}

// mean((a - b)^2) in two deterministic passes (griffinlim_iter's reconstruction error, griffinlim.py:268-269):
// per-workgroup partial sums in float64, then one workgroup adds them in a fixed order.
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_sqdiff_partial_kernel(const float *a, const float *b, int64_t n, double *part) {
    __shared__ double red[AP_BLOCK];
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const double d = (double)a[e] - (double)b[e];
        acc += d * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int sft = AP_BLOCK / 2; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_sum_partials_kernel(const double *part, int n_part, double scale, float *out) {
    __shared__ double red[AP_BLOCK];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_part; i += AP_BLOCK) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int sft = AP_BLOCK / 2; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * scale);
}

// per-workgroup max of x -> one integer atomic per workgroup on *key (order-preserving key)
AP_KERNEL void __launch_bounds__(AP_BLOCK) ap_reduce_max_kernel(const float *x, int64_t n, unsigned *key) {
    float *red = reinterpret_cast<float *>(ap_smem);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float m = -INFINITY;
    // 16-byte loads over the aligned body, scalar loads for the unaligned head and the tail
    const int64_t head = ((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) >> 2;
    const int64_t h = head < n ? head : n;
    const int64_t nq = (n - h) >> 2;
    const ap_float4 *xq = reinterpret_cast<const ap_float4 *>(x + h);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nq; e += stride) {
        const ap_float4 v = xq[e];
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    if (blockIdx.x == 0) {
        if ((int64_t)threadIdx.x < h) m = fmaxf(m, x[threadIdx.x]);
        const int64_t t = h + 4 * nq + threadIdx.x;
        if (t < n) m = fmaxf(m, x[t]);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) ap_atomic_max_u32(key, ap_fkey(red[0]));
}

// pass 1 of _to_db (convert.py:42-54): out = coef * log10(max(S, amin) / max(ref, amin));
// ref is *ref_key (max key of a previous reduction) when ref_key != NULL, else ref_value.
// top_db clip (convert.py:56-58: out = max(out, GLOBAL max(out) - top_db)): the conversion is
// monotone, so max(out) = dB(max(S)) and the floor is known from a read-only max reduction of S
// (*smax_key) before this single read+write pass.
struct ApDbParams {
    float coef, amin, ref_value, top_db;     // top_db < 0: no clip
    const unsigned *ref_key, *smax_key;
};
AP_DEV float ap_db_ref(const ApDbParams &D) {
    return fmaxf(D.ref_key ? ap_fkey_inv(*D.ref_key) : D.ref_value, D.amin);
}
AP_DEV float ap_db_value(const ApDbParams &D, float ref, float s) {
    return D.coef * log10f(fmaxf(s, D.amin) / ref);
}
AP_DEV float ap_db_floor(const ApDbParams &D, float ref) {
    return D.top_db >= 0.0f ? ap_db_value(D, ref, ap_fkey_inv(*D.smax_key)) - D.top_db : -INFINITY;
}

AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_to_db_kernel(const float *S, int64_t n, ApDbParams D, float *out) {
    const float ref = ap_db_ref(D);
    const float floor_v = ap_db_floor(D, ref);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride)
        out[e] = fmaxf(ap_db_value(D, ref, S[e]), floor_v);
}


// db_to_power / db_to_amplitude (convert.py:100-129, 169-198): ref * 10^(x / div)
AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_from_db_kernel(const float *x, int64_t n, float ref, float div, float *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride)
        out[e] = ref * powf(10.0f, x[e] / div);
}

// DCT-II (any small dense basis) along the middle axis of x viewed as (outer, n_in, inner):
//   out[o, k, i] = row_scale[k] * sum_m C[k, m] * x'[o, m, i]        (mfcc.py:135, :277-282)
// x' = x, or with DB = 1 the dB conversion (+ top_db clip) of x applied on the fly, so mfcc
// (mfcc.py:253-262) reads the mel power once and never writes the dB array.
// One thread per (o, i), all KT outputs of a chunk in registers; the basis sits transposed in
// LDS ([m][KT], 16-byte broadcast reads) and the loads of 8 input rows are issued together.
// WIDE = 0: x and out hold < 2^31 elements, so a thread's offset is one 32-bit register beside a
// row pointer the whole wave shares (global loads with a scalar base); WIDE = 1: 64-bit offsets.
// A reference level of exactly 1 (power_to_db's default, the only one mfcc uses) skips the division
// S / ref: dividing by 1 changes no bit.
template <int KT, int DB, int WIDE, int MINB = 1>
__global__ void __launch_bounds__(AP_BLOCK, MINB)
ap_dct_kernel(const float *__restrict__ x, const float *__restrict__ C, const float *__restrict__ row_scale,
              int64_t outer, int n_in, int64_t inner, int n_out, ApDbParams D, float *__restrict__ out) {
    typedef typename std::conditional<WIDE != 0, int64_t, uint32_t>::type off_t;
    float *Ct = reinterpret_cast<float *>(ap_smem);                 // [n_in][KT]
    const int64_t total = outer * inner;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float ref = 1.0f, floor_v = 0.0f;
    if (DB) { ref = ap_db_ref(D); floor_v = ap_db_floor(D, ref); }
    const bool unit_ref = !DB || ref == 1.0f;
    for (int k0 = 0; k0 < n_out; k0 += KT) {
        AP_LDS_BARRIER();
        for (int i = threadIdx.x; i < n_in * KT; i += AP_BLOCK) {
            const int m = i / KT, k = i - m * KT;
            Ct[i] = k0 + k < n_out ? C[(int64_t)(k0 + k) * n_in + m] : 0.0f;
        }
        AP_LDS_BARRIER();
        auto sweep = [&](auto unit) {                               // the whole pass, once per kind of reference level
            auto level = [&](float s) -> float {
                if (!DB) return s;
                const float c = fmaxf(s, D.amin);
                return fmaxf(D.coef * log10f(decltype(unit)::value ? c : c / ref), floor_v);
            };
            for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
                const int64_t o = e / inner, i = e - o * inner;
                const off_t xo = (off_t)(o * n_in * inner + i), oo = (off_t)(o * n_out * inner + i);
                float acc[KT];
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] = 0.0f;
                const float *row = x;                               // row m of every clip, wave-uniform
                auto load8 = [&](float (&d)[8]) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = (row + j * inner)[xo];
                    row += 8 * inner;
                };
                auto use8 = [&](const float (&d)[8], int m) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float vv = level(d[j]);
                        const float *c = Ct + (m + j) * KT;
#pragma unroll
                        for (int k = 0; k < KT; ++k) acc[k] = fmaf(c[k], vv, acc[k]);
                    }
                };
                // groups of 8 rows, the loads of the next group in flight under the arithmetic of this one
                const int groups = n_in >> 3;
                float cur[8], nxt[8];
                if (groups) load8(cur);
#pragma nounroll
                for (int g = 0; g < groups; ++g) {
                    if (g + 1 < groups) load8(nxt);
                    use8(cur, 8 * g);
#pragma unroll
                    for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
                }
                for (int m0 = 8 * groups; m0 < n_in; ++m0) {
                    const float vv = level(row[xo]);
                    row += inner;
                    const float *c = Ct + m0 * KT;
#pragma unroll
                    for (int k = 0; k < KT; ++k) acc[k] = fmaf(c[k], vv, acc[k]);
                }
                float *orow = out + (int64_t)k0 * inner;
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    if (k0 + k < n_out) orow[oo] = row_scale ? acc[k] * row_scale[k0 + k] : acc[k];
                    orow += inner;
                }
            }
        };
        if (unit_ref) sweep(std::true_type{});
        else sweep(std::false_type{});
    }
}

// any n_in (basis read through the scalar cache): fallback when the transposed basis does not fit LDS
template <int KT>
__global__ void __launch_bounds__(AP_BLOCK)
ap_dct_generic_kernel(const float *x, const float *C, const float *row_scale, int64_t outer, int n_in,
                      int64_t inner, int n_out, float *out) {
    const int64_t total = outer * inner;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t o = e / inner, i = e - o * inner;
        const float *xp = x + o * n_in * inner + i;
        float *op = out + o * n_out * inner + i;
        for (int k0 = 0; k0 < n_out; k0 += KT) {
            float acc[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) acc[k] = 0.0f;
            for (int m = 0; m < n_in; ++m) {
                const float v = xp[(int64_t)m * inner];
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k0 + k < n_out) acc[k] = fmaf(C[(int64_t)(k0 + k) * n_in + m], v, acc[k]);
            }
#pragma unroll
            for (int k = 0; k < KT; ++k)
                if (k0 + k < n_out)
                    op[(int64_t)(k0 + k) * inner] = row_scale ? acc[k] * row_scale[k0 + k] : acc[k];
        }
    }
}

// NumPy's default_rng(seed).uniform(low, high, n).astype(float32), element for element, on the
// device (reference griffinlim.py:112-115 draws the initial phase with the host Generator).
// PCG64 = 128-bit LCG (setseq) + XSL-RR output; one 64-bit output per double.  Element i is
// produced from the state advanced by i+1 steps (O(log i) jump-ahead, pcg_advance_lcg_128), so
// every lane is independent and the stores are coalesced.  float64 mul and add are kept
// un-fused to match NumPy's C arithmetic (low + range * next_double) bit for bit.
typedef unsigned __int128 ap_u128;

AP_DEV ap_u128 ap_pcg64_advance(ap_u128 state, ap_u128 inc, unsigned long long delta) {
    const ap_u128 MULT = ((ap_u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    ap_u128 cur_mult = MULT, cur_plus = inc, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    return acc_mult * state + acc_plus;
}

// the affine map of `delta` LCG steps: state_{k+delta} = A state_k + C
AP_DEV void ap_pcg64_affine(ap_u128 inc, unsigned long long delta, ap_u128 &A, ap_u128 &C) {
    const ap_u128 MULT = ((ap_u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    ap_u128 cur_mult = MULT, cur_plus = inc;
    A = 1;
    C = 0;
    while (delta > 0) {
        if (delta & 1) {
            A *= cur_mult;
            C = C * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
}

AP_KERNEL void __launch_bounds__(AP_BLOCK)
ap_pcg64_uniform_kernel(unsigned long long st_hi, unsigned long long st_lo, unsigned long long inc_hi,
                        unsigned long long inc_lo, double low, double range, int64_t n, float *out) {
#ifndef AP_HOST_EMU
#pragma clang fp contract(off)
#endif
    const ap_u128 state0 = ((ap_u128)st_hi << 64) | st_lo;
    const ap_u128 inc = ((ap_u128)inc_hi << 64) | inc_lo;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e0 >= n) return;
    // one O(log) jump to the thread's first element, then the affine map of `stride` LCG steps
    // (s -> A s + C, built once) from element to element: one 128-bit multiply-add per output
    ap_u128 s = ap_pcg64_advance(state0, inc, (unsigned long long)e0 + 1ULL);
    ap_u128 A, C;
    ap_pcg64_affine(inc, (unsigned long long)stride, A, C);
    for (int64_t e = e0; e < n; e += stride, s = A * s + C) {
        const unsigned long long hi = (unsigned long long)(s >> 64), lo = (unsigned long long)s;
        const unsigned long long x = hi ^ lo;
        const unsigned rot = (unsigned)(hi >> 58);
        const unsigned long long r = (x >> rot) | (x << ((64u - rot) & 63u));      // XSL-RR
        const double d = (double)(r >> 11) * (1.0 / 9007199254740992.0);
        const double v = low + range * d;
        out[e] = (float)v;
    }
}
