// Shared plain-C++ definitions: kernel parameter blocks, the FFT plan and launch
// geometry.  Included by the HIP translation unit and by the CPU test emulator
// (tests/emu), so nothing here may depend on the HIP runtime.
#pragma once
#include <stdint.h>

#define AP_MAX_PASSES 16
#define AP_BLOCK 256               // threads per workgroup of the LDS engine (4 wave64)
#define AP_LDS_TILE_BUDGET (64 * 1024)   // target LDS per workgroup when batching frames
#define AP_LDS_MAX (160 * 1024)    // gfx950 LDS per CU / max per workgroup
#define AP_MAX_G 16                // max frames per workgroup tile

// One complex float.  Under hipcc it is a clang ext-vector so that complex adds and multiplies
// map onto gfx950's packed-f32 VALU ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on an
// aligned VGPR pair = one complex); the CPU emulator (g++) sees the same layout as a struct.
#if defined(__clang__) && !defined(AP_HOST_EMU)
typedef float ap_float2 __attribute__((ext_vector_type(2)));
#define AP_PACKED_COMPLEX 1
#else
struct __attribute__((aligned(8))) ap_float2 {
    float x, y;
};
#endif

// 16-byte LDS / global accesses
struct __attribute__((aligned(16))) ap_float4 { float x, y, z, w; };
struct __attribute__((aligned(16))) ap_int4 { int x, y, z, w; };

// Radix plan of the complex transform that backs an n_fft-point real transform.
//   even n_fft : nc = n_fft/2 complex points (pack x[2n] + i x[2n+1]) + split pass
//   odd  n_fft : nc = n_fft complex points with zero imaginary part
struct ApFftPlan {
    int n;        // n_fft
    int nc;       // complex length
    int even;     // 1 if n even
    int tw_step;  // table stride turning W_nc^m into the W_n table index (2 if even else 1)
    int n_pass;
    int radix[AP_MAX_PASSES];
};

// Factor nc into radices this engine has in-register butterflies for (16,8,4,2,5,3)
// plus arbitrary primes (handled by the O(p) per-output generic pass).
static inline int ap_make_plan(int n_fft, ApFftPlan *p) {
    if (n_fft < 1) return -1;
    p->n = n_fft;
    p->even = (n_fft % 2 == 0) ? 1 : 0;
    p->nc = p->even ? n_fft / 2 : n_fft;
    p->tw_step = p->even ? 2 : 1;
    p->n_pass = 0;
    int m = p->nc;
    // powers of two: prefer 16s and 8s, finish with 4 / 2
    int e2 = 0;
    while (m % 2 == 0) { m /= 2; e2++; }
    while (e2 >= 4 && e2 != 5) { p->radix[p->n_pass++] = 16; e2 -= 4; }   // leave 5 = 8*4
    while (e2 >= 3) { p->radix[p->n_pass++] = 8; e2 -= 3; }
    if (e2 == 2) { p->radix[p->n_pass++] = 4; e2 = 0; }
    if (e2 == 1) { p->radix[p->n_pass++] = 2; e2 = 0; }
    while (m % 5 == 0) { p->radix[p->n_pass++] = 5; m /= 5; }
    while (m % 3 == 0) { p->radix[p->n_pass++] = 3; m /= 3; }
    for (int f = 7; (int64_t)f * f <= m; f += 2) {
        while (m % f == 0) {
            if (p->n_pass >= AP_MAX_PASSES) return -1;
            p->radix[p->n_pass++] = f;
            m /= f;
        }
    }
    if (m > 1) {
        if (p->n_pass >= AP_MAX_PASSES) return -1;
        p->radix[p->n_pass++] = m;
    }
    if (p->n_pass > AP_MAX_PASSES) return -1;
    return 0;
}

// Plan of a plain complex transform of length nc (used by the large-N four-step FFT).
static inline int ap_make_cplan(int nc, ApFftPlan *p) {
    if (nc < 1) return -1;
    int rc = ap_make_plan(2 * nc, p);      // even real length 2*nc -> complex length nc, same radices
    if (rc != 0) return rc;
    p->n = nc;
    p->even = 0;
    p->tw_step = 1;                        // the twiddle table is W_nc^j itself
    return 0;
}

// N = N1 * N2 with both factors <= AP_CFFT_MAX (LDS-resident); N2 = 1 when N fits.  -1 if impossible.
#define AP_CFFT_MAX 4096
static inline int ap_cfft_split(int64_t N, int *N1, int *N2) {
    if (N < 1) return -1;
    if (N <= AP_CFFT_MAX) { *N1 = (int)N; *N2 = 1; return 0; }
    int best = -1;
    int a0 = 2;
    while ((int64_t)a0 * a0 < N) ++a0;                                   // ceil(sqrt(N))
    for (int a = a0; a <= AP_CFFT_MAX; ++a) {                            // most balanced admissible split
        if (N % a == 0 && N / a <= AP_CFFT_MAX) { best = a; break; }
    }
    if (best < 0) return -1;
    *N1 = best;
    *N2 = (int)(N / best);
    return 0;
}

// LDS geometry of the generic engine: two ping-pong complex buffers per frame,
// `fstride` float2 per buffer per frame (padded by one to break the power-of-two
// stride between frames), G frames per workgroup.
struct ApTile {
    int G;          // frames per workgroup
    int fstride;    // float2 elements per frame per buffer
    int lds_bytes;  // dynamic LDS bytes
};

static inline int ap_make_tile(const ApFftPlan *p, int64_t T, ApTile *t) {
    int fstride = p->nc + 1;
    // the mel epilogue needs nc+1 floats of |X|^p per frame in the idle buffer: fits (2*fstride floats)
    int64_t per_frame = (int64_t)2 * fstride * (int64_t)sizeof(ap_float2);
    if (per_frame > AP_LDS_MAX) return -1;
    int G = (int)(AP_LDS_TILE_BUDGET / per_frame);
    if (G < 1) G = 1;
    if (G > AP_MAX_G) G = AP_MAX_G;
    if ((int64_t)G > T) G = (int)(T > 0 ? T : 1);
    t->G = G;
    t->fstride = fstride;
    t->lds_bytes = (int)(per_frame * G);
    return 0;
}

// Strided batched complex FFT (one leg of the four-step transform, kernels_bigfft.h)
struct ApCfftParams {
    const void *in;        // ap_float2 (or float when real_in)
    void *out;             // ap_float2 (or float when real_out)
    int64_t in_batch, out_batch;     // elements between signals of the batch
    int64_t in_fs, in_is;            // element strides: frame, index within frame
    int64_t out_fs, out_is;
    int64_t n_frames;
    int64_t tw_N;                    // > 0: multiply output (frame f, bin k) by W_tw_N^(f*k) (forward sign)
    int64_t tiles_per_signal;
    const ap_float2 *tw;             // W_n^j table of this leg
    int conj_io;                     // 1: conjugate input and output (inverse transform)
    int real_in, real_out;
    float scale;
    ApFftPlan plan;
    ApTile tile;
};

// Parameters of the fused framing + window + real FFT kernel and its epilogues.
struct ApStftParams {
    const float *y;        // (B, L)
    const float *window;   // (n_fft)
    const ap_float2 *tw;   // (n_fft) (cos, sin)(2 pi j / n_fft)
    int64_t L;
    int64_t T;
    int64_t tiles_per_clip;
    int64_t n_clips;
    int hop;
    int pad;               // left padding in samples (n_fft/2 if center else 0)
    int pad_mode;
    int n_bins;            // n_fft/2 + 1
    ApFftPlan plan;
    ApTile tile;
    // epilogue: complex spectrum
    ap_float2 *out_c;      // (B, F, T)
    // epilogue: mel
    float *out_mel;        // (B, M, T)
    const float *fb;       // (M, F)
    const int32_t *band_lo;
    const int32_t *band_len;
    int n_mels;
    float power;
    // contraction plan (ap_mel_plan_host): parts / weight quads / slot range of every row
    const int32_t *parts;
    const float *quads;
    const int32_t *rowstart;
    int n_parts, n_quads;
};

struct ApIrfftParams {
    const ap_float2 *S;    // (B, F, T)
    const ap_float2 *tw;
    float *frames;         // (B, T, n_fft)
    int64_t T;
    int64_t tiles_per_clip;
    int n_bins;
    ApFftPlan plan;
    ApTile tile;
};

static inline int64_t ap_n_frames(int64_t L, int n_fft, int hop, int center) {
    int64_t Lp = L + (center ? 2 * (int64_t)(n_fft / 2) : 0);
    if (Lp < n_fft) return 0;
    return 1 + (Lp - n_fft) / hop;
}
