// Compile-time specialisations of the LDS Stockham engine for the common small transforms
// (n_fft = 400 Whisper, 512, 1024): same data flow as ap_stft_generic_kernel, but the radix
// plan, the complex length NC and the tile height G are template constants (all index
// div/mod fold to shifts and multiplies), the window and the twiddle table are staged in LDS
// once per workgroup, and constant padding comes from the bounds-checked clip buffer instead
// of per-sample branches.  Workgroups are persistent over tiles.
#pragma once
#include "kernels_generic.h"
#include "kernels_wave.h"     // ApClip / ap_clip_load / ap_float4

// LDS layouts of a frame (ap_launch.h: ap_ct_fs).  The register-resident first pass writes its
// outputs TRANSPOSED: element idx = R0 j + q sits at q PQ + j, PQ = (NC / R0) | 1, so the R0
// stores of a thread hit consecutive lanes' slots (conflict-free) instead of a stride-R0 comb,
// and the second pass, which reads consecutive idx, walks it with the odd stride PQ
// (conflict-free as well).  Every later buffer is in natural order.
template <int TR0, int PQ>
AP_DEV constexpr int ap_ct_slot(int idx) { return TR0 > 0 ? (idx % TR0) * PQ + idx / TR0 : idx; }

// TR0 > 0: the source buffer is the transposed first-pass output of radix TR0.
// Twiddles W^(k i), i = 1..R-1, come from ONE table read (W^k) and powers of it: the engine is
// LDS-bound and the table reads of a radix-5 pass were as many as its data reads.
template <int R, int NC, int NS, int G, int FS, int TR0, int PQ, int NT>
AP_DEV void ap_stockham_pass_ct(const ap_float2 *in, ap_float2 *out, const ap_float2 *twl, int tid) {
    constexpr int PER = NC / R;
    constexpr int TMUL = (NC / (NS * R)) * 2;          // W_{NS*R}^1 in the W_n table (n = 2 NC)
    for (int item = tid; item < G * PER; item += NT) {
        // frames fastest across lanes: G is a power of two (shifts instead of divisions by PER),
        // and the frame stride FS = 4 (mod 32) spreads the 8 frames over the banks
        const int g = item % G;
        const int j = item / G;
        const ap_float2 *src = in + g * FS;
        ap_float2 *dst = out + g * FS;
        const int k = j % NS;
        ap_float2 v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = src[ap_ct_slot<TR0, PQ>(j + i * PER)];
        if (NS > 1) {
            ap_float2 w[R];
            w[1] = twl[TMUL * k];
#pragma unroll
            for (int i = 2; i < R; ++i) w[i] = (i & 1) ? ap_cmul(w[i - 1], w[1]) : ap_cmul(w[i / 2], w[i / 2]);
#pragma unroll
            for (int i = 1; i < R; ++i) v[i] = ap_mul_fw(v[i], w[i]);
        }
        ApButterfly<R>::run(v);
        const int j0 = (j / NS) * NS * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) dst[j0 + q * NS] = v[q];
    }
}

// ap_rfft_split (fft_lds.h) with packed arithmetic
AP_DEV ap_float2 ap_rfft_split_ct(const ap_float2 *Z, int nc, int k, const ap_float2 *tw) {
    const ap_float2 zk = Z[k == nc ? 0 : k];
    const ap_float2 zm = Z[k == 0 ? 0 : nc - k];
    const ap_float2 a = ap_add_conj(zk, zm), d = ap_sub_conj(zk, zm);
    const ap_float2 w = tw[k];   // (cos, sin)(2 pi k / n)
    // X[k] = a/2 - (i/2) W^k d
    const ap_float2 u = ap_mul_fw(d, ap_scale(w, 0.5f));
    return ap_fma_add_mi(a, ap_mk(0.5f, 0.5f), u);
}

// Both bins of a mirrored pair from one pair of reads (k in [0, nc/2]): X[k] and conj X[nc-k]
//   a = Z[k] + conj Z[nc-k], d = Z[k] - conj Z[nc-k], u = (W^k / 2) d
//   X[k] = a/2 + (-i) u,  conj X[nc-k] = a/2 - (-i) u          (k = 0: X[0] and X[nc])
AP_DEV void ap_rfft_split_pair_ct(const ap_float2 *Z, int nc, int k, const ap_float2 *tw, ap_float2 &xk,
                                  ap_float2 &xm_conj) {
    const ap_float2 zk = Z[k];
    const ap_float2 zm = Z[k == 0 ? 0 : nc - k];
    const ap_float2 a = ap_add_conj(zk, zm), d = ap_sub_conj(zk, zm);
    const ap_float2 u = ap_mul_fw(d, ap_scale(tw[k], 0.5f));
    const ap_float2 h = ap_mk(0.5f, 0.5f);
    xk = ap_fma_add_mi(a, h, u);
    xm_conj = ap_fma_sub_mi(a, h, u);
}

// EPI 0: complex (B,F,T); EPI 1: mel (B,M,T).  R2 = 1: two passes only.
// NT threads per workgroup (256 everywhere: 320 for the 8.5.5 plan, whose radix-5 passes have
// 8 x 40 = 320 butterflies per tile, measured slower - 0.31 against 0.20 ms - five waves do not
// spread over four SIMDs).
template <int EPI, int NC, int R0, int R1, int R2, int G, int PADGEN, int NT>
__global__ void __launch_bounds__(NT) ap_stft_ct_kernel(ApStftParams P) {
    constexpr int N = 2 * NC;
    constexpr int F = NC + 1;
    constexpr int PER0 = NC / R0;
    constexpr int PQ = PER0 | 1;                                        // odd row stride of the transposed buffer
    constexpr int FSMIN = R0 * PQ > NC + 1 ? R0 * PQ : NC + 1;
    constexpr int FS = ((FSMIN + 27) / 32) * 32 + 4;                    // = ap_ct_fs(N, R0)
    ap_float2 *bufA = reinterpret_cast<ap_float2 *>(ap_smem);
    ap_float2 *bufB = bufA + G * FS;
    ap_float2 *twl = bufB + G * FS;                    // [N] (cos, sin)(2 pi j / N)
    // mel plan tables (EPI 1 with a parts plan): weight quads, part descriptors, partial sums, row slots
    ap_float4 *wql = reinterpret_cast<ap_float4 *>(twl + N);
    ap_int4 *partl = reinterpret_cast<ap_int4 *>(wql + P.n_quads);
    float *partial = reinterpret_cast<float *>(partl + P.n_parts);            // [n_parts][G]
    int *rsl = reinterpret_cast<int *>(partial + P.n_parts * G);              // [M+1]
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += NT) twl[i] = P.tw[i];
    if (EPI == 1 && P.n_parts > 0) {
        for (int i = tid; i < P.n_quads; i += NT) wql[i] = reinterpret_cast<const ap_float4 *>(P.quads)[i];
        for (int i = tid; i < P.n_parts; i += NT) partl[i] = reinterpret_cast<const ap_int4 *>(P.parts)[i];
        for (int i = tid; i <= P.n_mels; i += NT) rsl[i] = P.rowstart[i];
    }
    AP_LDS_BARRIER();

    const int64_t n_tiles = P.tiles_per_clip * P.n_clips;
    // The first Stockham pass runs straight from registers: thread (g1, j1) owns butterfly j1 of
    // frame g1 of every tile, so it fetches exactly the R0 sample pairs c = j1 + i PER0 that
    // butterfly needs (one tile ahead: the HBM latency of tile i+1 hides behind the transform of
    // tile i) and keeps their window values in registers for the whole kernel.  No window table,
    // no windowed copy of the tile in LDS, one barrier less.
    static_assert(G * PER0 <= NT, "one first-pass butterfly per thread");
    const bool p1 = tid < G * PER0;
    const int g1 = tid / PER0, j1 = tid - g1 * PER0;
    ap_float2 wreg[R0], raw[R0];
#pragma unroll
    for (int i = 0; i < R0; ++i)
        wreg[i] = p1 ? reinterpret_cast<const ap_float2 *>(P.window)[j1 + i * PER0] : ap_mk(0.0f, 0.0f);
    auto load_tile = [&](int64_t tile) {
        const int64_t b = tile / P.tiles_per_clip;
        const int64_t t0 = (tile - b * P.tiles_per_clip) * G;
        const float *yb = P.y + b * P.L;
        const ApClip clip = ap_clip_make(yb, P.L);
        const bool live = p1 && t0 + g1 < P.T;
        const int64_t base = (t0 + g1) * (int64_t)P.hop - P.pad;
#pragma unroll
        for (int i = 0; i < R0; ++i) {
            const int64_t p = base + 2 * (j1 + i * PER0);
            raw[i] = ap_mk(0.0f, 0.0f);
            if (live) {
                if (PADGEN)
                    raw[i] = ap_mk(ap_load_padded(yb, P.L, p, P.pad_mode), ap_load_padded(yb, P.L, p + 1, P.pad_mode));
                else
                    raw[i] = ap_clip_load2(clip, (int)p);
            }
        }
    };
    if ((int64_t)blockIdx.x < n_tiles) load_tile(blockIdx.x);

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b = tile / P.tiles_per_clip;
        const int64_t t0 = (tile - b * P.tiles_per_clip) * G;
        const int Gt = (int)((P.T - t0) < G ? (P.T - t0) : G);

        // ---- window, first pass (NS = 1: no twiddles) in registers -> bufB ----------------------
        {
            ap_float2 v[R0];
#pragma unroll
            for (int i = 0; i < R0; ++i) v[i] = ap_mul2(wreg[i], raw[i]);
#ifndef AP_HOST_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (tile + gridDim.x < n_tiles) load_tile(tile + gridDim.x);
#ifndef AP_HOST_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
            ApButterfly<R0>::run(v);
            if (p1) {
                ap_float2 *dst = bufB + g1 * FS;
#pragma unroll
                for (int q = 0; q < R0; ++q) dst[q * PQ + j1] = v[q];
            }
        }
        AP_LDS_BARRIER();
        ap_stockham_pass_ct<R1, NC, R0, G, FS, R0, PQ, NT>(bufB, bufA, twl, tid);
        AP_LDS_BARRIER();
        ap_float2 *Z = bufA;
        ap_float2 *other = bufB;
        if (R2 > 1) {
            ap_stockham_pass_ct<(R2 > 1 ? R2 : 2), NC, R0 * R1, G, FS, 0, 1, NT>(bufA, bufB, twl, tid);
            AP_LDS_BARRIER();
            Z = bufB;
            other = bufA;
        }

        if (EPI == 0) {
            for (int item = tid; item < (NC / 2 + 1) * G; item += NT) {
                const int k = item / G;
                const int g = item - k * G;
                ap_float2 xk, xm;
                ap_rfft_split_pair_ct(Z + g * FS, NC, k, twl, xk, xm);
                if (g < Gt) {
                    P.out_c[(b * F + k) * P.T + t0 + g] = xk;
                    if (2 * k != NC) P.out_c[(b * F + (NC - k)) * P.T + t0 + g] = ap_mk(xm.x, -xm.y);
                }
            }
            AP_LDS_BARRIER();
        } else {
            float *Pw = reinterpret_cast<float *>(other);
            constexpr int PS = ((F + 3 + 27) / 32) * 32 + 4;   // = ap_ct_ps(N): 16-byte aligned planes, >= F + 3
            static_assert(PS <= 2 * FS, "power planes alias the idle exchange buffer");
            for (int item = tid; item < (NC / 2 + 1) * G; item += NT) {
                const int k = item / G;
                const int g = item - k * G;
                ap_float2 xk, xm;
                ap_rfft_split_pair_ct(Z + g * FS, NC, k, twl, xk, xm);     // |conj X| = |X|
                Pw[g * PS + k] = ap_pow_mag(xk.x, xk.y, P.power);
                if (2 * k != NC) Pw[g * PS + NC - k] = ap_pow_mag(xm.x, xm.y, P.power);
            }
            if (P.n_parts > 0) {
                // zero the alignment tail of every plane: the last weight quad may reach past bin F-1
                for (int item = tid; item < 3 * G; item += NT) {
                    const int g = item / 3, k = F + (item - g * 3);
                    if (k < PS) Pw[g * PS + k] = 0.0f;
                }
            }
            AP_LDS_BARRIER();
            if (P.n_parts > 0) {
                // plan-based banded contraction, everything from LDS (16-byte reads)
                for (int item = tid; item < P.n_parts * G; item += NT) {
                    const int p = item / G;
                    const int g = item - p * G;
                    const ap_int4 pd = partl[p];                       // slot, g0, ng, q0
                    const ap_float4 *pq = reinterpret_cast<const ap_float4 *>(Pw + g * PS) + pd.y;
                    const ap_float4 *wq = wql + pd.w;
                    float acc = 0.0f;
                    for (int i = 0; i < pd.z; ++i) {
                        const ap_float4 w = wq[i], q = pq[i];
                        acc = fmaf(w.x, q.x, acc);
                        acc = fmaf(w.y, q.y, acc);
                        acc = fmaf(w.z, q.z, acc);
                        acc = fmaf(w.w, q.w, acc);
                    }
                    partial[pd.x * G + g] = acc;
                }
                AP_LDS_BARRIER();
                for (int item = tid; item < P.n_mels * G; item += NT) {
                    const int m = item / G;
                    const int g = item - m * G;
                    if (g < Gt) {
                        float sum = 0.0f;
                        for (int j = rsl[m]; j < rsl[m + 1]; ++j) sum += partial[j * G + g];
                        P.out_mel[(b * P.n_mels + m) * P.T + t0 + g] = sum;
                    }
                }
            } else {
                for (int item = tid; item < P.n_mels * G; item += NT) {
                    const int m = item / G;
                    const int g = item - m * G;
                    if (g < Gt) {
                        const int lo = P.band_lo ? P.band_lo[m] : 0;
                        const int len = P.band_len ? P.band_len[m] : F;
                        const float *w = P.fb + (int64_t)m * F + lo;
                        const float *pp = Pw + g * PS + lo;
                        float acc = 0.0f;
                        for (int i = 0; i < len; ++i) acc = fmaf(w[i], pp[i], acc);
                        P.out_mel[(b * P.n_mels + m) * P.T + t0 + g] = acc;
                    }
                }
            }
            AP_LDS_BARRIER();
        }
    }
}
