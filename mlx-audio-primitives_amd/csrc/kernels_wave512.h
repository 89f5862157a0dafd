// Wave-per-frame fused mel-spectrogram for n_fft = 1024 on gfx950 (constant padding).
//
// The complex 512-point transform is 8 x 8 x 8 with 8 values per lane: three in-register
// radix-8 passes around two wave-private LDS transposes whose layouts are conflict-free for
// both the ds_write_b64 and the ds_read_b128 side (tools/lds_banks.py):
//
//   pass 1 over j (z[l + 64 j]), * W_512^(l k1)   -> T1: row 8 k1 + a of 10 complex, column b
//   pass 2 over b (l = a + 8 b),  * W_64^(a c)    -> T2: 66 c + 8 k1 + a
//   pass 3 over a                                  -> lane l = k1 + 8 c holds bins l + 64 e
//
// so after pass 3 every lane already owns the bins k = l + 64 r, r = 0..3, of the paired real
// split and only their mirrors 512 - k (upper half, natural order) come back from LDS.
// With 8 values per lane the kernel needs 147 VGPRs: 12 waves per CU (the 2048 kernel: 8; 16 waves
// under 128 VGPRs spilled inside the frame loop and measured 0.28 against 0.24 ms), and the
// 8-frame output run of a lane's two mel rows lives in registers (each lane stores 32 contiguous
// bytes per row), so a wave needs only its 5 KB exchange buffer in LDS.
// Everything after the split - power, plan-based contraction, row sums - is the scheme of
// kernels_wave.h.  Reference: mel.py:245-352 (stft.py:130 + mel.py:344-350).
#pragma once
#include <type_traits>
#include "kernels_wave.h"

#define APH_NC 512            // complex points
#define APH_WAVES 12          // waves per workgroup (3 per SIMD)
#define APH_X_COMPLEX 648     // exchange buffer: T1 64 x 10 = 640, T2 66 x 7 + 64 = 526, plane 516 floats
#define APH_PASSES 3          // contraction passes whose descriptors live in registers (192 entries)
#define APH_T1(r) ((r) * 10)
#define APH_T2(c, k1) ((c) * 66 + (k1) * 8)

// cos/sin(2 pi r / 16): W_1024^(64 r)
#define APH_C16(r) ((float)__builtin_cos(6.283185307179586476925 * (r) / 16.0))
#define APH_S16(r) ((float)__builtin_sin(6.283185307179586476925 * (r) / 16.0))

// windowed samples v[j] = z[lane + 64 j] -> v[e] = Z[lane + 64 e] (three radix-8 passes, two
// wave-private LDS transposes in X)
AP_DEV void aph_forward(ap_float2 (&v)[8], ap_float2 *X, const ap_float2 *TW1, const ap_float2 *TW2, int lane) {
    const int la = lane & 7, lb = lane >> 3;
    {   // pass 1 over j, twiddle W_512^(l k1), transpose #1
        ap_float2 t1[8];
#pragma unroll
        for (int k = 1; k < 8; ++k) t1[k] = TW1[k * 64 + lane];
        ApButterfly<8>::run(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = ap_mul_fw(v[k], t1[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) X[APH_T1(8 * k + la) + lb] = v[k];
    }
    AP_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = X[APH_T1(lane) + i];                  // lane = (k1, a), register b
    AP_WAVE_SYNC();
    {   // pass 2 over b, twiddle W_64^(a c), transpose #2
        ap_float2 t2[8];
#pragma unroll
        for (int c = 1; c < 8; ++c) t2[c] = TW2[la * 8 + c];
        ApButterfly<8>::run(v);
#pragma unroll
        for (int c = 1; c < 8; ++c) v[c] = ap_mul_fw(v[c], t2[c]);
#pragma unroll
        for (int c = 0; c < 8; ++c) X[APH_T2(c, lb) + la] = v[c];            // writer lane = (k1 = lb, a = la)
    }
    AP_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = X[APH_T2(lb, la) + i];                 // reader lane = (c = lb, k1 = la)
    AP_WAVE_SYNC();
    // pass 3 over a: this lane is (c = lb, k1 = la) and now holds Z[k1 + 8 c + 64 e] = Z[lane + 64 e]
    ApButterfly<8>::run(v);
}

// paired real split of Z (v[e] = Z[lane + 64 e]): xk[r] = X[lane + 64 r], xm[r] = X[512 - lane - 64 r]
// (CONJ = true) or its conjugate (enough for |X|), z256 = Z[256] (X[256] = conj Z[256]).  Only the
// mirrored upper half goes through LDS (natural order in X[0..256)).
template <bool CONJ>
AP_DEV void aph_split(const ap_float2 (&v)[8], ap_float2 *X, ap_float2 tws0h, int lane, ap_float2 (&xk)[4],
                      ap_float2 (&xm)[4], ap_float2 &z256) {
#pragma unroll
    for (int e = 4; e < 8; ++e) X[lane + 64 * (e - 4)] = v[e];                // Z[256 + lane + 64 (e-4)]
    AP_WAVE_SYNC();
    ap_float2 zm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int km = (APH_NC - (lane + 64 * r)) & (APH_NC - 1);             // mirror bin
        zm[r] = km >= 256 ? X[km - 256] : v[0];                               // km = 0 only for lane 0, r = 0
    }
    z256 = X[0];
    AP_WAVE_SYNC();
    const ap_float2 half = ap_mk(0.5f, 0.5f), halfc = ap_mk(0.5f, -0.5f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const ap_float2 a = ap_add_conj(v[r], zm[r]);
        const ap_float2 d = ap_sub_conj(v[r], zm[r]);
        const ap_float2 w = r == 0 ? tws0h : ap_mul_bw_c(tws0h, APH_C16(r), APH_S16(r));
        const ap_float2 u = ap_mul_fw(d, w);
        xk[r] = ap_fma_add_mi(a, half, u);
        xm[r] = CONJ ? ap_fma_sub_swap(a, halfc, u) : ap_fma_sub_mi(a, half, u);
    }
}

struct ApMelWave512Params {
    const float *y;            // (B, L)
    const float *window;       // (1024)
    const ap_float2 *tw;       // (1024) (cos, sin)(2 pi j / 1024)
    const int32_t *parts;      // wave layout of the plan (include/audioprims.h, desc[11])
    const float *quads;
    const int32_t *rowstart;
    float *out;                // (B, M, T)
    unsigned *max_key;
    int64_t L, T, n_clips;
    int64_t Ts;                // floats between the rows of `out` (T = dense; a multiple of 8 = whole sectors)
    int hop, pad, pad_mode, n_mels, n_parts, n_quads, n_slots, max_row_parts, partial_stride, hopj;
    float power;
    int off_tw1, off_tw2, off_win, off_wq, off_parts, off_partial, lds_bytes;
};

// HOPJ = 2: hop = 256, the next frame reuses 6 of this frame's 8 sample pairs per lane - with the loop trip of four
// frames and one copy of the body per register rotation of kernels_mel2048.h (no copies); HOPJ = 0: any hop, every
// frame loaded whole.  (clip, frame) advance incrementally; the 8-frame output run sits at static register positions.
// PADGEN: reflect / edge padding and odd hops - frames that reach over a clip end load through the index remap.
template <int PMODE, int HOPJ, int PADGEN = 0>
__global__ void __launch_bounds__(64 * APH_WAVES, 3) ap_mel1024_wave_kernel(ApMelWave512Params P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APH_X_COMPLEX;
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);   // [8][64]
    const ap_float2 *TW2 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw2);   // [8][8]
    const ap_float2 *WIN = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);   // [512] pairs
    const ap_float4 *WQ = reinterpret_cast<const ap_float4 *>(ap_smem + P.off_wq);
    const ap_int4 *PART = reinterpret_cast<const ap_int4 *>(ap_smem + P.off_parts);
    float *partial = reinterpret_cast<float *>(ap_smem + P.off_partial) + wave * P.partial_stride;
    const int M = P.n_mels;
    {   // workgroup tables (once; the only workgroup barrier)
        const int nt = 64 * APH_WAVES;
        ap_float2 *tw1 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1);
        ap_float2 *tw2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2);
        ap_float2 *win = reinterpret_cast<ap_float2 *>(ap_smem + P.off_win);
        ap_float4 *wq = reinterpret_cast<ap_float4 *>(ap_smem + P.off_wq);
        for (int i = tid; i < 8 * 64; i += nt) tw1[i] = P.tw[(2 * (i & 63) * (i >> 6)) & 1023];   // W_512^(l k1)
        if (tid < 64) tw2[tid] = P.tw[16 * (tid >> 3) * (tid & 7)];                                // W_64^(a c)
        for (int i = tid; i < APH_NC; i += nt) win[i] = reinterpret_cast<const ap_float2 *>(P.window)[i];
        for (int i = tid; i < P.n_quads; i += nt) wq[i] = reinterpret_cast<const ap_float4 *>(P.quads)[i];
        if (P.n_parts > 64 * APH_PASSES) {
            ap_int4 *part = reinterpret_cast<ap_int4 *>(ap_smem + P.off_parts);
            for (int i = tid; i < P.n_parts; i += nt) part[i] = reinterpret_cast<const ap_int4 *>(P.parts)[i];
        }
    }
    const ap_float2 tws0h = ap_scale(P.tw[lane], 0.5f);                      // W_1024^lane / 2
    float *pp = reinterpret_cast<float *>(X);                                 // |X|^p plane, aliased on X
    ap_int4 mypart[APH_PASSES];
#pragma unroll
    for (int ps = 0; ps < APH_PASSES; ++ps) {
        const int pi = lane + 64 * ps;
        mypart[ps].x = 0; mypart[ps].y = 0; mypart[ps].z = 0; mypart[ps].w = 0;
        if (pi < P.n_parts) mypart[ps] = reinterpret_cast<const ap_int4 *>(P.parts)[pi];
    }
    int rs0[2], rs1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = lane + 64 * i;
        rs0[i] = row < M ? P.rowstart[row] : 0;
        rs1[i] = row < M ? P.rowstart[row + 1] : 0;
    }
    AP_LDS_BARRIER();

    const int64_t worker = (int64_t)blockIdx.x * APH_WAVES + wave;
    const int64_t n_workers = (int64_t)gridDim.x * APH_WAVES;
    const int64_t n_frames = P.n_clips * P.T;
    const int64_t f_lo = n_frames * worker / n_workers, f_hi = n_frames * (worker + 1) / n_workers;
    float vmax = -INFINITY;
    if (f_lo < f_hi) {
        int64_t b = f_lo / P.T;                   // the only division: (clip, frame) advance incrementally
        int t = (int)(f_lo - b * P.T);
        const int Ti = (int)P.T;
        ApClip clip = ap_clip_make(P.y + b * P.L, P.L);
        constexpr int U = HOPJ == 2 ? 4 : 1;
        ap_float2 raw[8];
        auto ld2 = [&](int64_t bb, int base, int p) -> ap_float2 {
            if (PADGEN && !(base >= 0 && (int64_t)base + 2 * APH_NC <= P.L)) {
                const float *yb = P.y + bb * P.L;
                return ap_mk(ap_load_padded(yb, P.L, p, P.pad_mode), ap_load_padded(yb, P.L, p + 1, P.pad_mode));
            }
            return ap_clip_load2(clip, p);
        };
        auto load_frame = [&](int tt, auto rot_tag) {
            constexpr int ROT = decltype(rot_tag)::value;
            const int base = tt * P.hop - P.pad;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[(j + HOPJ * ROT) & 7] = ld2(b, base, base + 2 * (lane + 64 * j));
        };
        // U = 4: the stretch's first frame takes the register rotation t mod 4 and the run half (t / 4) mod 2, so frame t
        // sits at position t mod 8 of the run registers and full runs are the clip's frames 8 k .. 8 k + 7: aligned
        // 32-byte pieces of their rows (kernels_mel2048.h; whole sectors when the rows are padded to a multiple of 8)
        const int r0 = U == 4 ? (t & 3) : 0;
        if (r0 == 1) load_frame(t, std::integral_constant<int, 1 % U>());
        else if (r0 == 2) load_frame(t, std::integral_constant<int, 2 % U>());
        else if (r0 == 3) load_frame(t, std::integral_constant<int, 3 % U>());
        else load_frame(t, std::integral_constant<int, 0>());
        float acc0[8], acc1[8];                   // the run's values of rows lane and lane + 64
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
        int nrun = 0, half = U == 4 ? ((t >> 2) & 1) : 0;
        int64_t f = f_lo;

        auto frame = [&](auto rot_tag) -> bool {
            constexpr int ROT = decltype(rot_tag)::value;
            constexpr int NROT = (ROT + 1) % U;
            ap_float2 v[8];
            {
                ap_float2 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = WIN[lane + 64 * j];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ap_mul2(raw[(j + HOPJ * ROT) & 7], w[j]);
            }
            const bool clip_ends = t + 1 == Ti;
            const bool more = f + 1 < f_hi;
#pragma unroll
            for (int j = 0; j < 8; ++j) AP_PIN(v[j]);         // the samples' registers are free for the prefetch
            AP_SCHED_FENCE();
            if (more) {
                if (clip_ends) clip = ap_clip_make(P.y + (b + 1) * P.L, P.L);
                const int base = (clip_ends ? 0 : t + 1) * P.hop - P.pad;
#pragma unroll
                for (int j = 8 - HOPJ; j < 8; ++j) raw[(j + HOPJ * NROT) & 7] = ld2(clip_ends ? b + 1 : b, base, base + 2 * (lane + 64 * j));
                if (HOPJ == 0 || clip_ends) {
#pragma unroll
                    for (int j = 0; j < 8 - HOPJ; ++j)
                        raw[(j + HOPJ * NROT) & 7] = ld2(clip_ends ? b + 1 : b, base, base + 2 * (lane + 64 * j));
                }
            }
            AP_SCHED_FENCE();
            aph_forward(v, X, TW1, TW2, lane);
            {
                ap_float2 xk[4], xm[4], z256;
                aph_split<false>(v, X, tws0h, lane, xk, xm, z256);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = lane + 64 * r;
                    pp[k] = apw_pow2x<PMODE>(xk[r].x, xk[r].y, P.power);
                    pp[APH_NC - k] = apw_pow2x<PMODE>(xm[r].x, xm[r].y, P.power);
                }
                if (lane == 0) pp[APH_NC / 2] = apw_pow2x<PMODE>(z256.x, z256.y, P.power);
            }
            AP_WAVE_SYNC();
            // ---- plan-based contraction (kernels_wave.h) --------------------------------------------
#pragma unroll
            for (int ps = 0; ps < APH_PASSES; ++ps) {
                if (64 * ps < P.n_parts) {
                    const ap_int4 pd = mypart[ps];
                    const ap_float4 *pqa = reinterpret_cast<const ap_float4 *>(pp) + pd.y;
                    const ap_float4 *pqb = reinterpret_cast<const ap_float4 *>(pp) + pd.w;
                    const ap_float4 *wq = WQ + 256 * ps + lane;
                    ap_float4 w[4], q[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = wq[64 * i];
                    q[0] = pqa[0]; q[1] = pqa[1]; q[2] = pqb[0]; q[3] = pqb[1];
                    float acc[2] = {0.0f, 0.0f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i >> 1] = fmaf(w[i].x, q[i].x, acc[i >> 1]);
                        acc[i >> 1] = fmaf(w[i].y, q[i].y, acc[i >> 1]);
                        acc[i >> 1] = fmaf(w[i].z, q[i].z, acc[i >> 1]);
                        acc[i >> 1] = fmaf(w[i].w, q[i].w, acc[i >> 1]);
                    }
                    partial[pd.x] = pd.z < 0 ? acc[0] + acc[1] : acc[0];
                    partial[pd.z < 0 ? P.n_slots : pd.z] = acc[1];
                }
            }
            for (int p0 = 64 * APH_PASSES; p0 < P.n_parts; p0 += 64) {
                const ap_int4 pd = PART[p0 + lane];
                const ap_float4 *pqa = reinterpret_cast<const ap_float4 *>(pp) + pd.y;
                const ap_float4 *pqb = reinterpret_cast<const ap_float4 *>(pp) + pd.w;
                const ap_float4 *wq = WQ + 4 * p0 + lane;
                float acc[2] = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const ap_float4 w = wq[64 * i], q = (i & 2) ? pqb[i & 1] : pqa[i & 1];
                    acc[i >> 1] = fmaf(w.x, q.x, acc[i >> 1]);
                    acc[i >> 1] = fmaf(w.y, q.y, acc[i >> 1]);
                    acc[i >> 1] = fmaf(w.z, q.z, acc[i >> 1]);
                    acc[i >> 1] = fmaf(w.w, q.w, acc[i >> 1]);
                }
                partial[pd.x] = pd.z < 0 ? acc[0] + acc[1] : acc[0];
                partial[pd.z < 0 ? P.n_slots : pd.z] = acc[1];
            }
            AP_WAVE_SYNC();
            // ---- row sums into the run's registers ---------------------------------------------------
            float sum2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int cnt = rs1[i] - rs0[i];
                const float p0 = partial[rs0[i]], p1 = partial[rs0[i] + 1], p2 = partial[rs0[i] + 2],
                            p3 = partial[rs0[i] + 3];
                float sum = cnt > 0 ? p0 : 0.0f;
                sum += cnt > 1 ? p1 : 0.0f;
                sum += cnt > 2 ? p2 : 0.0f;
                sum += cnt > 3 ? p3 : 0.0f;
                if (P.max_row_parts > 4)
                    for (int j = rs0[i] + 4; j < rs1[i]; ++j) sum += partial[j];
                sum2[i] = sum;
                if (lane + 64 * i < M) vmax = fmaxf(vmax, sum);
            }
            // the frame's position in the run registers (kernels_mel2048.h)
            int pos;
            if (U == 1) {
#pragma unroll
                for (int i = 0; i < 7; ++i) { acc0[i] = acc0[i + 1]; acc1[i] = acc1[i + 1]; }
                acc0[7] = sum2[0];
                acc1[7] = sum2[1];
                pos = 7;
            } else {
                if (half) { acc0[4 + ROT] = sum2[0]; acc1[4 + ROT] = sum2[1]; }
                else { acc0[ROT] = sum2[0]; acc1[ROT] = sum2[1]; }
                pos = 4 * half + ROT;
            }
            ++nrun;
            AP_WAVE_SYNC();
            // ---- store the run when it is full, the clip ends or the stretch ends ---------------------
            if ((U == 1 ? nrun == 8 : pos == 7) || clip_ends || !more) {
                const int first = pos - nrun + 1;
                float *ob = P.out + b * (int64_t)M * P.Ts + (t - nrun + 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = lane + 64 * i;
                    if (row < M) {
                        float *dst = ob + (int64_t)row * P.Ts;
                        const float *src = i == 0 ? acc0 : acc1;
                        if (nrun == 8) {            // 32 contiguous bytes: two 16-byte stores (4-byte aligned)
                            ap_rsp_f4u lo, hi;
                            lo.x = src[0]; lo.y = src[1]; lo.z = src[2]; lo.w = src[3];
                            hi.x = src[4]; hi.y = src[5]; hi.z = src[6]; hi.w = src[7];
                            *reinterpret_cast<ap_rsp_f4u *>(dst) = lo;
                            *reinterpret_cast<ap_rsp_f4u *>(dst + 4) = hi;
                        } else {
#pragma unroll
                            for (int g = 0; g < 8; ++g)
                                if (g >= first && g <= pos) dst[g - first] = src[g];
                        }
                    }
                }
                nrun = 0;
            }
            if (clip_ends) { t = 0; ++b; } else { ++t; }
            ++f;
            return more;
        };
        if (U == 1) {
            while (frame(std::integral_constant<int, 0>())) {}
        } else {
            int skip = r0;                        // the first trip enters at rotation r0 (uniform branches)
            for (;;) {
                if (skip <= 0 && !frame(std::integral_constant<int, 0>())) break;
                if (skip <= 1 && !frame(std::integral_constant<int, 1 % U>())) break;
                if (skip <= 2 && !frame(std::integral_constant<int, 2 % U>())) break;
                if (!frame(std::integral_constant<int, 3 % U>())) break;
                skip = 0;
                half ^= 1;
            }
        }
    }
    if (P.max_key) {
        AP_WAVE_SYNC();
        partial[lane] = vmax;
        AP_WAVE_SYNC();
        if (lane == 0) {
            float m = partial[0];
            for (int i = 1; i < 64; ++i) m = fmaxf(m, partial[i]);
            ap_atomic_max_u32(P.max_key, ap_fkey(m));
        }
    }
}


// ---------------------------------------------------------------------------------------
// STFT, n_fft = 1024, complex output (B, 513, T), constant padding.  The scheme of
// ap_stft2048_wave_kernel (kernels_wave.h): the 8 waves of a workgroup transform 8 consecutive
// frames, transpose through LDS (2 chunks of 257 bins, one buffer: a wave needs 5 KB here, so two
// workgroups fit a CU and one stores while the other transforms) and write every row in
// sector-aligned 8-frame windows with register carries along a contiguous stretch of groups.
#define APHS_WAVES 8
#define APHS_OB_ROW 9
#define APHS_OB_ROWS 257

struct ApStftWave512Params {
    const float *y;            // (B, L)
    const float *window;       // (1024)
    const ap_float2 *tw;       // (1024)
    ap_float2 *out;            // (B, 513, T)
    int64_t L, T, groups_per_clip, n_groups;
    int hop, pad, pad_mode, padgen;   // padgen: reflect / edge padding or odd hops - index remap for the frames at the clip ends
    int off_tw1, off_tw2, off_win, off_ob, lds_bytes;
};

template <int PADGEN>
__global__ void __launch_bounds__(64 * APHS_WAVES, 4) ap_stft1024_wave_kernel(ApStftWave512Params P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APH_X_COMPLEX;
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);
    const ap_float2 *TW2 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw2);
    const ap_float2 *WIN = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);
    ap_float2 *OB = reinterpret_cast<ap_float2 *>(ap_smem + P.off_ob);        // [257][9]
    {
        const int nt = 64 * APHS_WAVES;
        ap_float2 *tw1 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1);
        ap_float2 *tw2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2);
        ap_float2 *win = reinterpret_cast<ap_float2 *>(ap_smem + P.off_win);
        for (int i = tid; i < 8 * 64; i += nt) tw1[i] = P.tw[(2 * (i & 63) * (i >> 6)) & 1023];
        if (tid < 64) tw2[tid] = P.tw[16 * (tid >> 3) * (tid & 7)];
        for (int i = tid; i < APH_NC; i += nt) win[i] = reinterpret_cast<const ap_float2 *>(P.window)[i];
    }
    const ap_float2 tws0h = ap_scale(P.tw[lane], 0.5f);
    AP_LDS_BARRIER();

    const int F = APH_NC + 1;
    const int sq = tid >> 3, sf = tid & 7;                                   // store role of this thread
    const int T7 = (int)(P.T & 7), Ti = (int)P.T;
    ap_float2 carry[2][2][2], carry_mid = ap_mk(0.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) carry[c][rr][0] = carry[c][rr][1] = ap_mk(0.0f, 0.0f);
    ap_float2 raw[8];
    auto load_frame = [&](int64_t group) {
        const int64_t b = group / P.groups_per_clip;
        const int64_t t = (group - b * P.groups_per_clip) * APHS_WAVES + wave;
        const ApClip clip = ap_clip_make(P.y + b * P.L, P.L);
        const int64_t base = t * (int64_t)P.hop - P.pad;
        // frames beyond T read past the clip: zeros (never stored)
        if (PADGEN && t < P.T && !(base >= 0 && base + 2 * APH_NC <= P.L)) {
            const float *yb = P.y + b * P.L;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t p = base + 2 * (lane + 64 * j);
                raw[j] = ap_mk(ap_load_padded(yb, P.L, p, P.pad_mode), ap_load_padded(yb, P.L, p + 1, P.pad_mode));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t p = base + 2 * (lane + 64 * j);
            raw[j] = ap_clip_load2(clip, (int)p);
        }
    };
    const int64_t g_lo = P.n_groups * (int64_t)blockIdx.x / gridDim.x;
    const int64_t g_hi = P.n_groups * ((int64_t)blockIdx.x + 1) / gridDim.x;
    if (g_lo < g_hi) load_frame(g_lo);

    for (int64_t group = g_lo; group < g_hi; ++group) {
        const int64_t b = group / P.groups_per_clip;
        const int64_t t0 = (group - b * P.groups_per_clip) * APHS_WAVES;
        ap_float2 *ob = P.out + b * (int64_t)F * P.T + t0;
        const int a0 = (int)(((reinterpret_cast<uintptr_t>(P.out) >> 3) + (uint64_t)(b * (int64_t)F * P.T + t0)) & 7);
        const bool have_prev = group > g_lo && t0 > 0;
        const bool last = group + 1 == g_hi || t0 + APHS_WAVES >= P.T;
        const int trem = (int)(P.T - t0);

        ap_float2 v[8];
        {
            ap_float2 w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = WIN[lane + 64 * j];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ap_mul2(raw[j], w[j]);
        }
        AP_SCHED_FENCE();
        if (group + 1 < g_hi) load_frame(group + 1);
        AP_SCHED_FENCE();
        aph_forward(v, X, TW1, TW2, lane);
        ap_float2 xk[4], xm[4], z256;
        aph_split<true>(v, X, tws0h, lane, xk, xm, z256);

        int k_rd[3];
        bool k_take[3], k_store[3], k_flush[3];
        ap_float2 *k_ob[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int ph = (0 - (a0 + (k == 0 ? sq * T7 : k == 1 ? -sq * T7 : 0))) & 7;
            k_rd[k] = (sf + ph) & 7;
            k_take[k] = sf < 8 - ph;
            const int dt = ph + sf - 8;
            k_store[k] = k_take[k] ? have_prev : dt < trem;
            k_flush[k] = k_take[k] && ph + sf < trem;
            k_ob[k] = ob + dt;
        }
        // ---- transposed store: chunk c holds r = 2c, 2c+1 (bins 64 r + lane and 512 - 64 r - lane) ----
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ap_float2 *buf = OB;
            if (c > 0) AP_LDS_BARRIER();                                     // chunk 0 fully read
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * c + rr;
                buf[(rr * 128 + lane) * APHS_OB_ROW + wave] = xk[r];
                buf[(rr * 128 + 64 + lane) * APHS_OB_ROW + wave] = xm[r];
            }
            if (c == 1 && lane == 0) buf[256 * APHS_OB_ROW + wave] = ap_mk(z256.x, -z256.y);   // X[256] = conj Z[256]
            AP_LDS_BARRIER();
            int bins[5], slots[5];
            ap_float2 x[5];
            const int ne = c == 1 ? 5 : 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * c + (i >> 1);
                bins[i] = (i & 1) ? APH_NC - 64 * r - sq : 64 * r + sq;
                slots[i] = (i >> 1) * 128 + (i & 1) * 64 + sq;
            }
            bins[4] = APH_NC / 2;
            slots[4] = 256;
            // phi(bin) = -(a0 + bin T) mod 8 only depends on the element's class (bins 64 r + sq: k = 0, bins
            // 512 - 64 r - sq: k = 1, bin 256: k = 2; 64 T, 512 T and 256 T are multiples of 8): computed once per group
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i < ne) x[i] = buf[slots[i] * APHS_OB_ROW + k_rd[i == 4 ? 2 : (i & 1)]];
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i < ne) {
                    const int k = i == 4 ? 2 : (i & 1);
                    ap_float2 &cy = i < 4 ? carry[c][i >> 1][i & 1] : carry_mid;
                    const bool mine = i < 4 || tid < APHS_WAVES;
                    const ap_float2 val = k_take[k] ? cy : x[i];
                    if (mine && k_store[k]) k_ob[k][bins[i] * Ti] = val;
                    if (k_take[k]) cy = x[i];
                }
            if (last) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < ne) {
                        const bool mine = i < 4 || tid < APHS_WAVES;
                        if (mine && k_flush[i == 4 ? 2 : (i & 1)]) k_ob[i == 4 ? 2 : (i & 1)][bins[i] * Ti + 8] = x[i];
                    }
            }
        }
        AP_LDS_BARRIER();                                                    // OB free for the next group
    }
}


// ---------------------------------------------------------------------------------------
// ISTFT, n_fft = 1024, hop 128 / 256 / 512: irfft of every frame + overlap-add in one kernel
// (stft.py:292-338), the scheme of ap_irfft2048_wave_kernel<1>.  The 8 waves of a workgroup take 8
// consecutive frames: the (bin, 8 frames) segments are staged through LDS (2 chunks of 257 bins,
// next group's rows prefetched into registers), Hermitian merge at half scale, the mirrored half
// exchanged through the wave's buffer, the 8 x 8 x 8 transform on conjugated data, then every wave
// leaves its windowed frame in its buffer and the workgroup gathers the 8 hop positions the group
// completes (<= n_fft / hop frames each, increasing frame order as overlap_add.metal:16-55),
// divides by the window-sum-of-squares and stores; later positions go to an LDS carry (ping-pong)
// for the next group of the workgroup's contiguous stretch.  A stretch that starts inside a clip
// first runs the preceding group with its stores disabled.
struct ApIstftWave512Params {
    const ap_float2 *S;        // (B, 513, T)
    const ap_float2 *tw;       // (1024)
    const float *window;       // (1024) synthesis window
    float *y;                  // (B, out_len)
    int64_t T, groups_per_clip, n_groups, out_offset, out_len;
    int hop;
    int off_tw1, off_tw2, off_win, off_ib, off_carry, lds_bytes;
};

__global__ void __launch_bounds__(64 * APHS_WAVES, 4) ap_istft1024_wave_kernel(ApIstftWave512Params P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APH_X_COMPLEX;
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);
    const ap_float2 *TW2 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw2);
    const ap_float2 *WINP = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);   // (w[2n], w[2n+1])
    const float *WIN = reinterpret_cast<const float *>(ap_smem + P.off_win);
    ap_float2 *IB = reinterpret_cast<ap_float2 *>(ap_smem + P.off_ib);                  // [257][9]
    {
        const int nt = 64 * APHS_WAVES;
        ap_float2 *tw1 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1);
        ap_float2 *tw2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2);
        float *win = reinterpret_cast<float *>(ap_smem + P.off_win);
        for (int i = tid; i < 8 * 64; i += nt) tw1[i] = P.tw[(2 * (i & 63) * (i >> 6)) & 1023];
        if (tid < 64) tw2[tid] = P.tw[16 * (tid >> 3) * (tid & 7)];
        for (int i = tid; i < 2 * APH_NC; i += nt) win[i] = P.window[i];
    }
    const ap_float2 tws0h = ap_scale(P.tw[lane], 0.5f);
    const ap_float2 half = ap_mk(0.5f, 0.5f), halfc = ap_mk(0.5f, -0.5f);
    AP_LDS_BARRIER();
    const int F = APH_NC + 1;
    const float scale = 1.0f / 512.0f;     // 1/n_fft, and the merge works at half scale
    const int H = P.hop;
    const int hs = H == 128 ? 7 : (H == 256 ? 8 : 9);
    const int CN = 2 * APH_NC - H;                                   // carry length
    const int Ti = (int)P.T;

    const int64_t g_lo = P.n_groups * (int64_t)blockIdx.x / gridDim.x;
    const int64_t g_hi = P.n_groups * ((int64_t)blockIdx.x + 1) / gridDim.x;
    // thread (sq = tid / 8, sf = tid % 8) fetches frame sf of rows sq, 64 + sq, 128 + sq, 192 + sq of both
    // chunks (bins 64 r + sq and 512 - 64 r - sq, r = 2c, 2c + 1), one group ahead, into registers
    const int sq = tid >> 3, sf = tid & 7;
    ap_float2 pre[2][4], pre_mid = ap_mk(0.0f, 0.0f);
    auto load_group = [&](int64_t group) {
        const int64_t b = group / P.groups_per_clip;
        const int64_t t0 = (group - b * P.groups_per_clip) * APHS_WAVES;
        const bool live = t0 + sf < P.T;
        const ap_float2 *sb = P.S + b * (int64_t)F * P.T + t0 + sf;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * c + (i >> 1);
                const int bin = (i & 1) ? APH_NC - 64 * r - sq : 64 * r + sq;
                pre[c][i] = live ? sb[bin * Ti] : ap_mk(0.0f, 0.0f);
            }
        if (tid < APHS_WAVES) pre_mid = live ? sb[(APH_NC / 2) * Ti] : ap_mk(0.0f, 0.0f);
    };
    const int64_t g_first = (g_lo < g_hi && g_lo % P.groups_per_clip != 0) ? g_lo - 1 : g_lo;
    if (g_first < g_hi) load_group(g_first);
    for (int64_t group = g_first; group < g_hi; ++group) {
        const int64_t b = group / P.groups_per_clip;
        const int64_t t0 = (group - b * P.groups_per_clip) * APHS_WAVES;

        // ---- transpose through LDS: every wave collects its frame's bins in registers -------
        ap_float2 xk[4], xm[4], xh = ap_mk(0.0f, 0.0f);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c > 0) AP_LDS_BARRIER();                             // chunk 0 fully read
#pragma unroll
            for (int i = 0; i < 4; ++i)
                IB[((i >> 1) * 128 + (i & 1) * 64 + sq) * APHS_OB_ROW + sf] = pre[c][i];
            if (c == 1 && tid < APHS_WAVES) IB[256 * APHS_OB_ROW + sf] = pre_mid;
            AP_LDS_BARRIER();
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                xk[2 * c + rr] = IB[(rr * 128 + lane) * APHS_OB_ROW + wave];
                xm[2 * c + rr] = IB[(rr * 128 + 64 + lane) * APHS_OB_ROW + wave];
            }
            if (c == 1) xh = IB[256 * APHS_OB_ROW + wave];
        }
        AP_SCHED_FENCE();
        if (group + 1 < g_hi) load_group(group + 1);
        AP_SCHED_FENCE();
        // ---- Hermitian merge: conj(Z[k]) / 2 and conj(Z[512-k]) / 2 of the packed inverse -----
        //   a = X[k] + conj X[512-k], d = X[k] - conj X[512-k], o = (W^-k / 2) d
        //   conj Z[k] / 2 = conj(a/2 + i o),  conj Z[512-k] / 2 = a/2 - i o
        ap_float2 v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ap_float2 a_k = xk[r], a_m = xm[r];
            if (r == 0 && lane == 0) { a_k.y = 0.0f; a_m.y = 0.0f; }      // DC / Nyquist imaginary parts ignored
            const ap_float2 a = ap_add_conj(a_k, a_m);
            const ap_float2 d = ap_sub_conj(a_k, a_m);
            const ap_float2 w = r == 0 ? tws0h : ap_mul_bw_c(tws0h, APH_C16(r), APH_S16(r));
            const ap_float2 o = ap_mul_bw(d, w);                           // W^-k = (c, +s)
            v[r] = ap_fma_sub_swap(a, halfc, o);                            // index lane + 64 r
            // index 512 - k belongs to the mirrored lane: upper half, natural order, through LDS
            const int km = (APH_NC - (lane + 64 * r)) & (APH_NC - 1);
            if (!(r == 0 && lane == 0)) X[km - 256] = ap_fma_add_mi(a, half, o);
        }
        if (lane == 0) X[0] = xh;           // bin 256 pairs with itself: conj Z[256] / 2 = X[256]
        AP_WAVE_SYNC();
#pragma unroll
        for (int e = 4; e < 8; ++e) v[e] = X[lane + 64 * (e - 4)];
        AP_WAVE_SYNC();
        aph_forward(v, X, TW1, TW2, lane);
        // y[n] = conj(Y[n]) / 1024 -> samples 2n, 2n+1 of the frame; n = lane + 64 e
        float *carry_in = reinterpret_cast<float *>(ap_smem + P.off_carry) + (int)(group & 1) * CN;
        float *carry_out = reinterpret_cast<float *>(ap_smem + P.off_carry) + (int)((group + 1) & 1) * CN;
        {   // windowed frame -> this wave's exchange buffer, natural order
            ap_float2 wv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wv[e] = WINP[lane + 64 * e];
#pragma unroll
            for (int e = 0; e < 8; ++e) X[lane + 64 * e] = ap_mul2(ap_mul2(v[e], ap_mk(scale, -scale)), wv[e]);
        }
        if (t0 == 0)
            for (int i = tid; i < CN; i += 64 * APHS_WAVES) carry_in[i] = 0.0f;
        AP_LDS_BARRIER();
        const bool emit = group >= g_lo;
        const bool clip_last = t0 + APHS_WAVES >= P.T;
        const float *XF = reinterpret_cast<const float *>(ap_smem);   // frame f at XF + f * 2 APH_X_COMPLEX
        const int n_own = APHS_WAVES * H;
        const int64_t p0 = t0 * (int64_t)H;
        const int t0i = (int)t0;
        float *yb = P.y + b * P.out_len;
        const int64_t n0 = p0 - P.out_offset;
        for (int r = 4 * tid; r < n_own + CN; r += 4 * 64 * APHS_WAVES) {
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            if (r < CN) {
                const ap_float4 c4 = *reinterpret_cast<const ap_float4 *>(carry_in + r);
                s0 = c4.x; s1 = c4.y; s2 = c4.z; s3 = c4.w;
            }
            const int f_cov = ((r - 2 * APH_NC) >> hs) + 1;
            const int f_lo = f_cov < 0 ? 0 : f_cov;
            int f_hi = r >> hs;
            if (f_hi > APHS_WAVES - 1) f_hi = APHS_WAVES - 1;
            for (int f = f_lo; f <= f_hi; ++f) {
                const ap_float4 q = *reinterpret_cast<const ap_float4 *>(XF + f * (2 * APH_X_COMPLEX) + (r - (f << hs)));
                s0 += q.x; s1 += q.y; s2 += q.z; s3 += q.w;
            }
            if (r >= n_own) {
                ap_float4 c4; c4.x = s0; c4.y = s1; c4.z = s2; c4.w = s3;
                *reinterpret_cast<ap_float4 *>(carry_out + (r - n_own)) = c4;
            }
            if (emit && (r < n_own || clip_last)) {
                int F_lo = f_cov < -t0i ? -t0i : f_cov;
                int F_hi = r >> hs;
                if (F_hi > Ti - 1 - t0i) F_hi = Ti - 1 - t0i;
                float w0 = 0.0f, w1 = 0.0f, w2 = 0.0f, w3 = 0.0f;
                for (int Fi = F_lo; Fi <= F_hi; ++Fi) {
                    const ap_float4 w = *reinterpret_cast<const ap_float4 *>(WIN + (r - Fi * H));
                    w0 += w.x * w.x; w1 += w.y * w.y; w2 += w.z * w.z; w3 += w.w * w.w;
                }
                const int64_t n = n0 + r;
                if (n >= 0 && n < P.out_len) yb[n] = s0 / fmaxf(w0, 1e-8f);
                if (n + 1 >= 0 && n + 1 < P.out_len) yb[n + 1] = s1 / fmaxf(w1, 1e-8f);
                if (n + 2 >= 0 && n + 2 < P.out_len) yb[n + 2] = s2 / fmaxf(w2, 1e-8f);
                if (n + 3 >= 0 && n + 3 < P.out_len) yb[n + 3] = s3 / fmaxf(w3, 1e-8f);
            }
        }
        if (emit && clip_last) {
            int64_t n = p0 + n_own + CN - P.out_offset;
            if (n < 0) n = 0;
            for (n += tid; n < P.out_len; n += 64 * APHS_WAVES) yb[n] = 0.0f;
        }
        AP_LDS_BARRIER();                  // frames and staging buffer free for the next group
    }
}
