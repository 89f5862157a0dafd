// LDS-resident Stockham FFT engine for gfx950: in-register radix-{2,3,4,5,8,16}
// butterflies, an O(p)-per-output pass for any other prime, and the real-input
// split / merge steps.  Device code only (also compiled by the CPU emulator in
// tests/emu through a shim that maps __device__/threadIdx/__syncthreads).
//
// Stands in for mx.fft.rfft / mx.fft.irfft (reference stft.py:130, :295), which
// live in the un-vendored mlx dependency.
#pragma once
// Non-template kernels of the kernel headers: one translation unit owns them with external linkage; every
// other unit that includes the header for its device helpers (AP_TU_SECONDARY) gets private copies.
#ifdef AP_TU_SECONDARY
#define AP_KERNEL static __global__
#else
#define AP_KERNEL __global__
#endif
#include "ap_common.h"

#ifndef AP_DEV
#define AP_DEV __device__ __forceinline__
#endif

// Workgroup barrier for kernels whose threads only ever communicate through LDS: wait for this
// wave's LDS traffic, then s_barrier.  __syncthreads() would also drain every outstanding global
// store (vmcnt(0), the workgroup-scope release) and expose the full HBM write latency at each
// barrier of a store-heavy kernel.
#ifdef AP_HOST_EMU
#define AP_LDS_BARRIER() __syncthreads()
#else
#define AP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// order-preserving float <-> uint key so a float max can use an integer atomic
AP_DEV unsigned ap_fkey(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
AP_DEV float ap_fkey_inv(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __builtin_bit_cast(float, u);
}

#ifdef AP_HOST_EMU
inline void ap_atomic_max_u32(unsigned *p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
#else
AP_DEV void ap_atomic_max_u32(unsigned *p, unsigned v) { atomicMax(p, v); }
#endif

// ---- complex arithmetic on one (re, im) register pair ---------------------------------
// Device build: packed-f32 instructions.  The compiler folds whole-vector negations and
// component swaps into VOP3P op_sel modifiers but not the half negations a multiply by -i or a
// conjugate needs, so those forms are spelled as one instruction each (op_sel picks the half
// of a source that feeds the low / high result lane, neg_lo / neg_hi negate it).
#ifdef AP_PACKED_COMPLEX
AP_DEV ap_float2 ap_mk(float x, float y) { ap_float2 r = {x, y}; return r; }
AP_DEV ap_float2 ap_add(ap_float2 a, ap_float2 b) { return a + b; }
AP_DEV ap_float2 ap_sub(ap_float2 a, ap_float2 b) { return a - b; }
AP_DEV ap_float2 ap_scale(ap_float2 a, float s) { return a * s; }
AP_DEV ap_float2 ap_mul2(ap_float2 a, ap_float2 b) { return a * b; }                  // per component
AP_DEV ap_float2 ap_fma_s(ap_float2 a, float s, ap_float2 b) {                         // a*s + b
    return __builtin_elementwise_fma(a, ap_mk(s, s), b);
}
AP_DEV ap_float2 ap_mul_mi(ap_float2 a) { return ap_mk(a.y, -a.x); }                  // -i a
AP_DEV ap_float2 ap_fma2(ap_float2 a, ap_float2 b, ap_float2 c) { return __builtin_elementwise_fma(a, b, c); }    // a*b + c
AP_DEV ap_float2 ap_fnma2(ap_float2 a, ap_float2 b, ap_float2 c) { return __builtin_elementwise_fma(a, -b, c); }  // c - a*b
#define AP_PK2(name, insn)                                                    \
    AP_DEV ap_float2 name(ap_float2 a, ap_float2 b) {                          \
        ap_float2 d;                                                           \
        asm(insn : "=v"(d) : "v"(a), "v"(b));                                  \
        return d;                                                              \
    }
#define AP_PK3(name, insn)                                                    \
    AP_DEV ap_float2 name(ap_float2 a, ap_float2 b, ap_float2 c) {            \
        ap_float2 d;                                                           \
        asm(insn : "=v"(d) : "v"(a), "v"(b), "v"(c));                          \
        return d;                                                              \
    }
AP_PK2(ap_add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")   // a + (-i) b
AP_PK2(ap_sub_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")   // a - (-i) b
AP_PK2(ap_add_conj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                              // a + conj b
AP_PK2(ap_sub_conj, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                              // a - conj b
// (b.y a.x... ) second half of a complex multiply: c + (a.y b.y, -a.x b.y)  /  c + (-a.y b.y, a.x b.y)
AP_PK3(ap_cmul_tail_fw, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")
AP_PK3(ap_cmul_tail_bw, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")
// a*h + (-i) u, a*h - (-i) u and a*h - (u.y, u.x)   (h applied per component)
AP_PK3(ap_fma_add_mi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]")
AP_PK3(ap_fma_sub_mi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1]")
AP_PK3(ap_fma_sub_swap, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")
// a * (c - i s): multiply by the forward twiddle whose table entry is (c, s)
AP_DEV ap_float2 ap_mul_fw(ap_float2 a, ap_float2 w) { return ap_cmul_tail_fw(a, w, a * w.xx); }
// a * (c + i s)
AP_DEV ap_float2 ap_mul_bw(ap_float2 a, ap_float2 w) { return ap_cmul_tail_bw(a, w, a * w.xx); }
// (c1, s1) (c2, s2) -> (cos, sin) of the summed angle: the product of two table-form twiddles
AP_DEV ap_float2 ap_cmul(ap_float2 a, ap_float2 b) { return ap_mul_bw(a, b); }
// compile-time twiddle: the compiler keeps (c, c) and (s, -s) as scalar register pairs
AP_DEV ap_float2 ap_mul_fw_c(ap_float2 a, float c, float s) {
    return __builtin_elementwise_fma(a.yx, ap_mk(s, -s), a * ap_mk(c, c));
}
AP_DEV ap_float2 ap_mul_bw_c(ap_float2 a, float c, float s) {
    return __builtin_elementwise_fma(a.yx, ap_mk(-s, s), a * ap_mk(c, c));
}
#else
AP_DEV ap_float2 ap_mk(float x, float y) { ap_float2 r; r.x = x; r.y = y; return r; }
AP_DEV ap_float2 ap_add(ap_float2 a, ap_float2 b) { return ap_mk(a.x + b.x, a.y + b.y); }
AP_DEV ap_float2 ap_sub(ap_float2 a, ap_float2 b) { return ap_mk(a.x - b.x, a.y - b.y); }
AP_DEV ap_float2 ap_scale(ap_float2 a, float s) { return ap_mk(a.x * s, a.y * s); }
AP_DEV ap_float2 ap_mul2(ap_float2 a, ap_float2 b) { return ap_mk(a.x * b.x, a.y * b.y); }
AP_DEV ap_float2 ap_fma_s(ap_float2 a, float s, ap_float2 b) { return ap_mk(a.x * s + b.x, a.y * s + b.y); }
AP_DEV ap_float2 ap_mul_mi(ap_float2 a) { return ap_mk(a.y, -a.x); }
AP_DEV ap_float2 ap_fma2(ap_float2 a, ap_float2 b, ap_float2 c) { return ap_mk(a.x * b.x + c.x, a.y * b.y + c.y); }
AP_DEV ap_float2 ap_fnma2(ap_float2 a, ap_float2 b, ap_float2 c) { return ap_mk(c.x - a.x * b.x, c.y - a.y * b.y); }
AP_DEV ap_float2 ap_add_mi(ap_float2 a, ap_float2 b) { return ap_mk(a.x + b.y, a.y - b.x); }
AP_DEV ap_float2 ap_sub_mi(ap_float2 a, ap_float2 b) { return ap_mk(a.x - b.y, a.y + b.x); }
AP_DEV ap_float2 ap_add_conj(ap_float2 a, ap_float2 b) { return ap_mk(a.x + b.x, a.y - b.y); }
AP_DEV ap_float2 ap_sub_conj(ap_float2 a, ap_float2 b) { return ap_mk(a.x - b.x, a.y + b.y); }
AP_DEV ap_float2 ap_cmul_tail_fw(ap_float2 a, ap_float2 w, ap_float2 c) { return ap_mk(c.x + a.y * w.y, c.y - a.x * w.y); }
AP_DEV ap_float2 ap_cmul_tail_bw(ap_float2 a, ap_float2 w, ap_float2 c) { return ap_mk(c.x - a.y * w.y, c.y + a.x * w.y); }
AP_DEV ap_float2 ap_fma_add_mi(ap_float2 a, ap_float2 h, ap_float2 u) { return ap_mk(a.x * h.x + u.y, a.y * h.y - u.x); }
AP_DEV ap_float2 ap_fma_sub_mi(ap_float2 a, ap_float2 h, ap_float2 u) { return ap_mk(a.x * h.x - u.y, a.y * h.y + u.x); }
AP_DEV ap_float2 ap_fma_sub_swap(ap_float2 a, ap_float2 h, ap_float2 u) { return ap_mk(a.x * h.x - u.y, a.y * h.y - u.x); }
AP_DEV ap_float2 ap_mul_fw(ap_float2 a, ap_float2 w) {
    return ap_mk(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
}
AP_DEV ap_float2 ap_mul_bw(ap_float2 a, ap_float2 w) {
    return ap_mk(a.x * w.x - a.y * w.y, a.y * w.x + a.x * w.y);
}
AP_DEV ap_float2 ap_cmul(ap_float2 a, ap_float2 b) { return ap_mul_bw(a, b); }
AP_DEV ap_float2 ap_mul_fw_c(ap_float2 a, float c, float s) { return ap_mul_fw(a, ap_mk(c, s)); }
AP_DEV ap_float2 ap_mul_bw_c(ap_float2 a, float c, float s) { return ap_mul_bw(a, ap_mk(c, s)); }
#endif

// ---- forward butterflies, natural-order outputs ---------------------------------
AP_DEV void ap_fft2(ap_float2 &a, ap_float2 &b) {
    ap_float2 t = ap_sub(a, b);
    a = ap_add(a, b);
    b = t;
}

AP_DEV void ap_fft4(ap_float2 &a0, ap_float2 &a1, ap_float2 &a2, ap_float2 &a3) {
    ap_float2 t0 = ap_add(a0, a2), t1 = ap_sub(a0, a2);
    ap_float2 t2 = ap_add(a1, a3), e = ap_sub(a1, a3);
    a0 = ap_add(t0, t2);
    a2 = ap_sub(t0, t2);
    a1 = ap_add_mi(t1, e);
    a3 = ap_sub_mi(t1, e);
}
// the same with input a2 still to be multiplied by -i (saves materialising the rotation)
AP_DEV void ap_fft4_a2mi(ap_float2 &a0, ap_float2 &a1, ap_float2 &a2, ap_float2 &a3) {
    ap_float2 t0 = ap_add_mi(a0, a2), t1 = ap_sub_mi(a0, a2);
    ap_float2 t2 = ap_add(a1, a3), e = ap_sub(a1, a3);
    a0 = ap_add(t0, t2);
    a2 = ap_sub(t0, t2);
    a1 = ap_add_mi(t1, e);
    a3 = ap_sub_mi(t1, e);
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
        ap_float2 m = ap_fma_s(t1, -0.5f, v[0]);
        ap_float2 s = ap_scale(t2, S3);
        v[0] = ap_add(v[0], t1);
        v[1] = ap_add_mi(m, s);
        v[2] = ap_sub_mi(m, s);
    }
};

template <>
struct ApButterfly<5> {
    static AP_DEV void run(ap_float2 *v) {
        const float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;
        const float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;
        ap_float2 a1 = ap_add(v[1], v[4]), a2 = ap_add(v[2], v[3]);
        ap_float2 b1 = ap_sub(v[1], v[4]), b2 = ap_sub(v[2], v[3]);
        ap_float2 m1 = ap_fma_s(a2, C2, ap_fma_s(a1, C1, v[0]));
        ap_float2 m2 = ap_fma_s(a2, C1, ap_fma_s(a1, C2, v[0]));
        ap_float2 n1 = ap_fma_s(b2, S2, ap_scale(b1, S1));
        ap_float2 n2 = ap_fma_s(b2, -S1, ap_scale(b1, S2));
        v[0] = ap_add(ap_add(v[0], a1), a2);
        // V1 = m1 - i n1, V4 = m1 + i n1, V2 = m2 - i n2, V3 = m2 + i n2
        v[1] = ap_add_mi(m1, n1);
        v[4] = ap_sub_mi(m1, n1);
        v[2] = ap_add_mi(m2, n2);
        v[3] = ap_sub_mi(m2, n2);
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
        o1 = ap_mul_fw_c(o1, H, H);
        o3 = ap_mul_fw_c(o3, -H, H);
        v[0] = ap_add(e0, o0); v[4] = ap_sub(e0, o0);
        v[1] = ap_add(e1, o1); v[5] = ap_sub(e1, o1);
        v[2] = ap_add_mi(e2, o2); v[6] = ap_sub_mi(e2, o2);
        v[3] = ap_add(e3, o3); v[7] = ap_sub(e3, o3);
    }
};

// radix-4 butterfly of the products x_i w_i (w applied per component: a window on packed real pairs):
// the first level's sums and differences take the second product as an FMA - 6 instructions for the 4
// multiplies and 4 additions
AP_DEV void ap_fft4_weighted(const ap_float2 &x0, const ap_float2 &w0, const ap_float2 &x1, const ap_float2 &w1,
                             const ap_float2 &x2, const ap_float2 &w2, const ap_float2 &x3, const ap_float2 &w3,
                             ap_float2 &a0, ap_float2 &a1, ap_float2 &a2, ap_float2 &a3) {
    const ap_float2 m0 = ap_mul2(x0, w0), m1 = ap_mul2(x1, w1);
    const ap_float2 t0 = ap_fma2(x2, w2, m0), t1 = ap_fnma2(x2, w2, m0);
    const ap_float2 t2 = ap_fma2(x3, w3, m1), e = ap_fnma2(x3, w3, m1);
    a0 = ap_add(t0, t2);
    a2 = ap_sub(t0, t2);
    a1 = ap_add_mi(t1, e);
    a3 = ap_sub_mi(t1, e);
}

template <>
struct ApButterfly<16> {
    static AP_DEV void run(ap_float2 *v) {
        ap_float2 a[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r][0] = v[r]; a[r][1] = v[r + 4]; a[r][2] = v[r + 8]; a[r][3] = v[r + 12];
            ap_fft4(a[r][0], a[r][1], a[r][2], a[r][3]);
        }
        finish(a, v);
    }
    // v = DFT16 of (x[i] w[i]), the weights folded into the first level
    static AP_DEV void run_weighted(const ap_float2 *x, const ap_float2 *w, ap_float2 *v) {
        ap_float2 a[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            ap_fft4_weighted(x[r], w[r], x[r + 4], w[r + 4], x[r + 8], w[r + 8], x[r + 12], w[r + 12],
                             a[r][0], a[r][1], a[r][2], a[r][3]);
        finish(a, v);
    }
    static AP_DEV void finish(ap_float2 (&a)[4][4], ap_float2 *v) {
        // cos/sin(2 pi m / 16), m = 1,2,3 (others by symmetry)
        const float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
        const float H = 0.70710678118654752440f;
        // twiddle a[r][p] *= W16^(r*p); entry (c,s) means c - i s
        // r=1: p=1 (C1,S1) p=2 (H,H) p=3 (S1,C1)
        a[1][1] = ap_mul_fw_c(a[1][1], C1, S1);
        a[1][2] = ap_mul_fw_c(a[1][2], H, H);
        a[1][3] = ap_mul_fw_c(a[1][3], S1, C1);
        // r=2: p=1 (H,H) p=2 (0,1) = -i, folded into the p=2 butterfly below, p=3 (-H,H)
        a[2][1] = ap_mul_fw_c(a[2][1], H, H);
        a[2][3] = ap_mul_fw_c(a[2][3], -H, H);
        // r=3: p=1 (S1,C1) p=2 (-H,H) p=3 m=9 -> (cos(9pi/8), sin(9pi/8)) = (-C1,-S1)
        a[3][1] = ap_mul_fw_c(a[3][1], S1, C1);
        a[3][2] = ap_mul_fw_c(a[3][2], -H, H);
        a[3][3] = ap_mul_fw_c(a[3][3], -C1, -S1);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p == 2) ap_fft4_a2mi(a[0][p], a[1][p], a[2][p], a[3][p]);
            else ap_fft4(a[0][p], a[1][p], a[2][p], a[3][p]);
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
        AP_LDS_BARRIER();
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
