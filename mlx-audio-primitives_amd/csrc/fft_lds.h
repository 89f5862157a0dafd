// LDS-resident Stockham FFT engine for gfx950: in-register radix-{2,3,4,5,8,16}
// butterflies, an O(p)-per-output pass for any other prime, and the real-input
// split / merge steps.  Device code only (also compiled by the CPU emulator in
// tests/emu through a shim that maps __device__/threadIdx/__syncthreads).
//
// Stands in for mx.fft.rfft / mx.fft.irfft (reference stft.py:130, :295), which
// live in the un-vendored mlx dependency.
#pragma once
#include "ap_common.h"

#ifndef AP_DEV
#define AP_DEV __device__ __forceinline__
#endif

AP_DEV ap_float2 ap_mk(float x, float y) { ap_float2 r; r.x = x; r.y = y; return r; }
AP_DEV ap_float2 ap_add(ap_float2 a, ap_float2 b) { return ap_mk(a.x + b.x, a.y + b.y); }
AP_DEV ap_float2 ap_sub(ap_float2 a, ap_float2 b) { return ap_mk(a.x - b.x, a.y - b.y); }
// a * (c - i s): multiply by the forward twiddle whose table entry is (c, s)
AP_DEV ap_float2 ap_mul_fw(ap_float2 a, ap_float2 w) {
    return ap_mk(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
}
AP_DEV ap_float2 ap_mul_mi(ap_float2 a) { return ap_mk(a.y, -a.x); }   // -i * a

// ---- forward butterflies, natural-order outputs ---------------------------------
AP_DEV void ap_fft2(ap_float2 &a, ap_float2 &b) {
    ap_float2 t = ap_sub(a, b);
    a = ap_add(a, b);
    b = t;
}

AP_DEV void ap_fft4(ap_float2 &a0, ap_float2 &a1, ap_float2 &a2, ap_float2 &a3) {
    ap_float2 t0 = ap_add(a0, a2), t1 = ap_sub(a0, a2);
    ap_float2 t2 = ap_add(a1, a3), t3 = ap_mul_mi(ap_sub(a1, a3));
    a0 = ap_add(t0, t2);
    a2 = ap_sub(t0, t2);
    a1 = ap_add(t1, t3);
    a3 = ap_sub(t1, t3);
}

template <int R>
struct ApButterfly;

template <>
struct ApButterfly<2> {
    static AP_DEV void run(ap_float2 *v) { ap_fft2(v[0], v[1]); }
};

template <>
struct ApButterfly<4> {
    static AP_DEV void run(ap_float2 *v) { ap_fft4(v[0], v[1], v[2], v[3]); }
};

template <>
struct ApButterfly<3> {
    static AP_DEV void run(ap_float2 *v) {
        const float S3 = 0.86602540378443864676f;
        ap_float2 t1 = ap_add(v[1], v[2]);
        ap_float2 t2 = ap_sub(v[1], v[2]);
        ap_float2 m = ap_mk(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
        ap_float2 s = ap_mk(S3 * t2.x, S3 * t2.y);
        v[0] = ap_add(v[0], t1);
        v[1] = ap_mk(m.x + s.y, m.y - s.x);
        v[2] = ap_mk(m.x - s.y, m.y + s.x);
    }
};

template <>
struct ApButterfly<5> {
    static AP_DEV void run(ap_float2 *v) {
        const float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;
        const float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;
        ap_float2 a1 = ap_add(v[1], v[4]), a2 = ap_add(v[2], v[3]);
        ap_float2 b1 = ap_sub(v[1], v[4]), b2 = ap_sub(v[2], v[3]);
        ap_float2 m1 = ap_mk(v[0].x + C1 * a1.x + C2 * a2.x, v[0].y + C1 * a1.y + C2 * a2.y);
        ap_float2 m2 = ap_mk(v[0].x + C2 * a1.x + C1 * a2.x, v[0].y + C2 * a1.y + C1 * a2.y);
        ap_float2 n1 = ap_mk(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
        ap_float2 n2 = ap_mk(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
        v[0] = ap_mk(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
        // V1 = m1 - i n1, V4 = m1 + i n1, V2 = m2 - i n2, V3 = m2 + i n2
        v[1] = ap_mk(m1.x + n1.y, m1.y - n1.x);
        v[4] = ap_mk(m1.x - n1.y, m1.y + n1.x);
        v[2] = ap_mk(m2.x + n2.y, m2.y - n2.x);
        v[3] = ap_mk(m2.x - n2.y, m2.y + n2.x);
    }
};

template <>
struct ApButterfly<8> {
    static AP_DEV void run(ap_float2 *v) {
        const float H = 0.70710678118654752440f;
        ap_float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        ap_float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
        ap_fft4(e0, e1, e2, e3);
        ap_fft4(o0, o1, o2, o3);
        // W8^1 = (1-i)/sqrt2, W8^2 = -i, W8^3 = (-1-i)/sqrt2
        o1 = ap_mk(H * (o1.x + o1.y), H * (o1.y - o1.x));
        o2 = ap_mul_mi(o2);
        o3 = ap_mk(H * (o3.y - o3.x), -H * (o3.x + o3.y));
        v[0] = ap_add(e0, o0); v[4] = ap_sub(e0, o0);
        v[1] = ap_add(e1, o1); v[5] = ap_sub(e1, o1);
        v[2] = ap_add(e2, o2); v[6] = ap_sub(e2, o2);
        v[3] = ap_add(e3, o3); v[7] = ap_sub(e3, o3);
    }
};

template <>
struct ApButterfly<16> {
    static AP_DEV void run(ap_float2 *v) {
        // cos/sin(2 pi m / 16), m = 1,2,3 (others by symmetry)
        const float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
        const float H = 0.70710678118654752440f;
        ap_float2 a[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r][0] = v[r]; a[r][1] = v[r + 4]; a[r][2] = v[r + 8]; a[r][3] = v[r + 12];
            ap_fft4(a[r][0], a[r][1], a[r][2], a[r][3]);
        }
        // twiddle a[r][p] *= W16^(r*p); entry (c,s) means c - i s
        // r=1: p=1 (C1,S1) p=2 (H,H) p=3 (S1,C1)
        a[1][1] = ap_mul_fw(a[1][1], ap_mk(C1, S1));
        a[1][2] = ap_mul_fw(a[1][2], ap_mk(H, H));
        a[1][3] = ap_mul_fw(a[1][3], ap_mk(S1, C1));
        // r=2: p=1 (H,H) p=2 (0,1) p=3 (-H,H)
        a[2][1] = ap_mul_fw(a[2][1], ap_mk(H, H));
        a[2][2] = ap_mul_mi(a[2][2]);
        a[2][3] = ap_mul_fw(a[2][3], ap_mk(-H, H));
        // r=3: p=1 (S1,C1) p=2 (-H,H) p=3 m=9 -> (cos(9pi/8), sin(9pi/8)) = (-C1,-S1)
        a[3][1] = ap_mul_fw(a[3][1], ap_mk(S1, C1));
        a[3][2] = ap_mul_fw(a[3][2], ap_mk(-H, H));
        a[3][3] = ap_mul_fw(a[3][3], ap_mk(-C1, -S1));
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ap_fft4(a[0][p], a[1][p], a[2][p], a[3][p]);
            v[p] = a[0][p]; v[p + 4] = a[1][p]; v[p + 8] = a[2][p]; v[p + 12] = a[3][p];
        }
    }
};

// ---- one Stockham pass over a tile of G frames held in LDS -------------------------
//  in/out : tile base; frame g lives at + g*fstride
//  Ns     : product of the radices of the earlier passes
//  forward transform: v[i] *= W_{Ns*R}^{k*i},  W = exp(-2 pi i / (Ns*R))
template <int R>
AP_DEV void ap_stockham_pass(const ap_float2 *in, ap_float2 *out, const ApFftPlan &pl, int Ns,
                             const ap_float2 *tw, int G, int fstride, int tid, int nthreads) {
    const int nc = pl.nc;
    const int per_frame = nc / R;
    const int total = G * per_frame;
    const int tmul = (nc / (Ns * R)) * pl.tw_step;   // table index of W_{Ns*R}^1
    for (int item = tid; item < total; item += nthreads) {
        const int g = item / per_frame;
        const int j = item - g * per_frame;
        const ap_float2 *src = in + g * fstride;
        ap_float2 *dst = out + g * fstride;
        const int k = j % Ns;
        ap_float2 v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = src[j + i * per_frame];
        if (Ns > 1) {
            const int tk = tmul * k;
#pragma unroll
            for (int i = 1; i < R; ++i) v[i] = ap_mul_fw(v[i], tw[tk * i]);
        }
        ApButterfly<R>::run(v);
        const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) dst[j0 + q * Ns] = v[q];
    }
}

// Any other prime radix p: one work item per output element, O(p) each.
AP_DEV void ap_stockham_pass_prime(const ap_float2 *in, ap_float2 *out, const ApFftPlan &pl,
                                   int p, int Ns, const ap_float2 *tw, int G, int fstride,
                                   int tid, int nthreads) {
    const int nc = pl.nc;
    const int per_frame = nc / p;
    const int total = G * nc;
    const int64_t e_tw = nc / (Ns * p);   // exponent of W_nc per unit of k*i
    const int64_t e_bf = nc / p;          // exponent of W_nc per unit of i*q
    for (int item = tid; item < total; item += nthreads) {
        const int g = item / nc;
        const int rem = item - g * nc;
        const int q = rem / per_frame;
        const int j = rem - q * per_frame;
        const ap_float2 *src = in + g * fstride;
        const int k = j % Ns;
        const int64_t step = ((int64_t)k * e_tw + (int64_t)q * e_bf) % nc;
        ap_float2 acc = src[j];
        int64_t e = 0;
        for (int i = 1; i < p; ++i) {
            e += step;
            if (e >= nc) e -= nc;
            acc = ap_add(acc, ap_mul_fw(src[j + i * per_frame], tw[e * pl.tw_step]));
        }
        const int j0 = (j / Ns) * Ns * p + k;
        out[g * fstride + j0 + q * Ns] = acc;
    }
}

// Run every pass of the plan.  Data starts in `a`; returns the buffer that holds the
// natural-order result (a or b).  Ends with a barrier.
AP_DEV ap_float2 *ap_fft_tile(ap_float2 *a, ap_float2 *b, const ApFftPlan &pl, const ap_float2 *tw,
                              int G, int fstride, int tid, int nthreads) {
    int Ns = 1;
    ap_float2 *src = a, *dst = b;
    for (int s = 0; s < pl.n_pass; ++s) {
        const int R = pl.radix[s];
        switch (R) {
            case 16: ap_stockham_pass<16>(src, dst, pl, Ns, tw, G, fstride, tid, nthreads); break;
            case 8: ap_stockham_pass<8>(src, dst, pl, Ns, tw, G, fstride, tid, nthreads); break;
            case 4: ap_stockham_pass<4>(src, dst, pl, Ns, tw, G, fstride, tid, nthreads); break;
            case 2: ap_stockham_pass<2>(src, dst, pl, Ns, tw, G, fstride, tid, nthreads); break;
            case 5: ap_stockham_pass<5>(src, dst, pl, Ns, tw, G, fstride, tid, nthreads); break;
            case 3: ap_stockham_pass<3>(src, dst, pl, Ns, tw, G, fstride, tid, nthreads); break;
            default: ap_stockham_pass_prime(src, dst, pl, R, Ns, tw, G, fstride, tid, nthreads); break;
        }
        Ns *= R;
        __syncthreads();
        ap_float2 *t = src; src = dst; dst = t;
    }
    return src;
}

// Forward real split (even n): X[k] from the nc-point spectrum Z of x[2n] + i x[2n+1].
//   X[k] = (Z[k] + conj Z[nc-k])/2 - (i/2) W_n^k (Z[k] - conj Z[nc-k]),  k in [0, nc]
AP_DEV ap_float2 ap_rfft_split(const ap_float2 *Z, int nc, int k, const ap_float2 *tw) {
    const ap_float2 zk = Z[k == nc ? 0 : k];
    const ap_float2 zm = Z[k == 0 ? 0 : nc - k];
    const float ax = 0.5f * (zk.x + zm.x), ay = 0.5f * (zk.y - zm.y);
    const float dx = 0.5f * (zk.x - zm.x), dy = 0.5f * (zk.y + zm.y);
    const ap_float2 w = tw[k];   // (cos, sin)(2 pi k / n)
    return ap_mk(ax + (w.x * dy - w.y * dx), ay - (w.x * dx + w.y * dy));
}
