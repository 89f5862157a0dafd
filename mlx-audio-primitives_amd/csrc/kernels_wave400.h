// Fused mel-spectrogram for n_fft = 400 (the Whisper front end: hop 160, 80 filters) on gfx950:
// one wave64 transforms EIGHT consecutive frames at a time, 8 lanes per frame, 25 complex values per
// lane, with no workgroup barrier and no LDS pass inside the transform.
//
// The packed 200-point complex transform z[m] = x[2m] + i x[2m+1] is 25 x 8:
//
//   lane (g = lane >> 3, q = lane & 7) holds z_g[q + 8 r], r = 0..24  (bounds-checked buffer loads:
//   the hardware range check is the constant padding; window pairs live in registers)
//   radix-25 in registers (5 x 5, compile-time W_25 twiddles)      [mx.fft.rfft, stft.py:130]
//   * W_200^(q k1) (small LDS table)
//   radix-8 ACROSS the 8 lanes of a frame: one ds_swizzle (lane ^ 4) stage, then the quad radix-4
//   with DPP quad_perm as in kernels_wave.h -> lane q owns the 25 contiguous bins
//   k = k1 + 25 k2(q), k2(q) = 2 bitrev2(q & 3) + (q >> 2)
//   paired real split: Z[200 - k] sits in lane 7 - q (DPP row_half_mirror), register 25 - k1
//   |X|^p -> the wave's 8 power planes in LDS
//   mel contraction: lane (g, q) owns frame g and the filters m = q + 8 i: a dot product over the
//   filter's band with zero-padded weights from an LDS table (the 8 lanes of a step hold 8 adjacent
//   filters, so their band lengths agree to within a few bins), four steps per trip so that the LDS
//   reads of 32 filters are in flight together; results go straight to HBM.
//
// Per frame ~160 wave-instructions against ~360 (and four workgroup barriers per 8 frames) in the
// compile-time LDS engine that served this shape before (kernels_ct.h).  Constant padding /
// center=False, n_mels <= 128, band lengths <= 32.  Reference: mel.py:245-352.
#pragma once
#include "kernels_wave.h"

#define APQ_WAVES 8          // waves per workgroup (they only share read-only LDS tables)
#define APQ_NC 200           // complex points
#define APQ_PS 264           // floats per power plane: 201 bins + zero tail for padded band reads, = 8 (mod 32)
#define APQ_WMAX 32          // floats per filter row of the LDS weight table

struct ApMel400Params {
    const float *y;            // (B, L)
    const float *window;       // (400)
    const ap_float2 *tw;       // (400) (cos, sin)(2 pi j / 400)
    const float *fb;           // (M, 201) dense filterbank
    const int32_t *band_lo, *band_len;   // (M) span of each filter's non-zeros
    float *out;                // (B, M, T)
    unsigned *max_key;
    int64_t L, T, n_clips, groups_per_clip, n_groups;
    int hop, pad, n_mels;
    float power;
    int off_t200, off_s400, off_w, off_lo, off_plane, lds_bytes;
};

#define APQ_C25(e) ((float)__builtin_cos(6.283185307179586476925 * (e) / 25.0))
#define APQ_S25(e) ((float)__builtin_sin(6.283185307179586476925 * (e) / 25.0))

#ifdef AP_HOST_EMU
AP_DEV float apq_xor4(float x) { return emu_lane_xor(x, 4); }
AP_DEV float apq_mirror8(float x) { return emu_lane_xor(x, 7); }
AP_DEV float apq_lane_read(float x, int src_lane) { return emu_lane_perm(x, src_lane); }
#else
AP_DEV float apq_xor4(float x) {      // value of lane ^ 4: ds_swizzle bit mode (and 0x1f, or 0, xor 4), no LDS memory
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x101F));
}
AP_DEV float apq_mirror8(float x) {   // value of lane ^ 7: DPP row_half_mirror
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
}
AP_DEV float apq_lane_read(float x, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, x)));
}
#endif

// 25-point DFT in registers: r = r0 + 5 r1 in, k1 = kappa1 + 5 kappa0 out
AP_DEV void apq_dft25(ap_float2 (&v)[25]) {
    ap_float2 a[5][5];
#pragma unroll
    for (int r0 = 0; r0 < 5; ++r0) {
        ap_float2 t[5];
#pragma unroll
        for (int r1 = 0; r1 < 5; ++r1) t[r1] = v[r0 + 5 * r1];
        ApButterfly<5>::run(t);
#pragma unroll
        for (int k = 0; k < 5; ++k)
            a[r0][k] = (r0 > 0 && k > 0) ? ap_mul_fw_c(t[k], APQ_C25(r0 * k), APQ_S25(r0 * k)) : t[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        ap_float2 u[5];
#pragma unroll
        for (int r0 = 0; r0 < 5; ++r0) u[r0] = a[r0][k];
        ApButterfly<5>::run(u);
#pragma unroll
        for (int k0 = 0; k0 < 5; ++k0) v[k + 5 * k0] = u[k0];
    }
}

template <int PMODE>
__global__ void __launch_bounds__(64 * APQ_WAVES, 2) ap_mel400_wave_kernel(ApMel400Params P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    const int g = lane >> 3, q = lane & 7;
    ap_float2 *T200 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_t200);      // [8][25] W_200^(q k1)
    ap_float2 *S400 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_s400);      // [8][25] W_400^(k1 + 25 k2(q)) / 2
    float *WT = reinterpret_cast<float *>(ap_smem + P.off_w);                   // [M8][APQ_WMAX] zero-padded band weights
    int *LO = reinterpret_cast<int *>(ap_smem + P.off_lo);                      // [M8] band start; [M8 + i]: longest band of step i
    float *plane = reinterpret_cast<float *>(ap_smem + P.off_plane) + wave * (8 * APQ_PS);
    const int M = P.n_mels;
    const int NI = (M + 7) / 8;                 // contraction steps: filters 8 i + q
    const int M8 = 8 * NI;

    // ---------------- workgroup tables (once; the only workgroup barrier) ----------------
    {
        const int nt = 64 * APQ_WAVES;
        for (int i = tid; i < 8 * 25; i += nt) {
            const int qq = i / 25, k1 = i - qq * 25;
            const int k2 = 2 * (((qq & 1) << 1) | ((qq >> 1) & 1)) + (qq >> 2);
            // W_200^(q k1) times the lane's own factor of the lane ^ 4 stage (sg = s1 s2 of the quad stage):
            // sg for q < 4 (the lane keeps sg u_q), -sg W_8^(q & 3) for q >= 4 (it keeps -sg W u_q)
            const int ja = qq & 3;
            const float sg = (ja == 1 || ja == 2) ? -1.0f : 1.0f;
            const ap_float2 t = P.tw[(2 * qq * k1) % 400];
            const ap_float2 f = qq < 4 ? ap_mk(sg, 0.0f) : ap_scale(P.tw[50 * ja], -sg);
            T200[i] = ap_mk(t.x * f.x - t.y * f.y, t.x * f.y + t.y * f.x);     // product of two forward-form twiddles
            S400[i] = ap_scale(P.tw[k1 + 25 * k2], 0.5f);
        }
        for (int i = tid; i < M8 * APQ_WMAX; i += nt) {
            const int m = i / APQ_WMAX, j = i - m * APQ_WMAX;
            WT[i] = (m < M && j < P.band_len[m]) ? P.fb[(int64_t)m * (APQ_NC + 1) + P.band_lo[m] + j] : 0.0f;
        }
        for (int i = tid; i < M8; i += nt) LO[i] = i < M ? P.band_lo[i] : 0;
        for (int i = tid; i < NI; i += nt) {
            int mx = 0;
            for (int j = 0; j < 8; ++j) {
                const int m = 8 * i + j;
                if (m < M && P.band_len[m] > mx) mx = P.band_len[m];
            }
            LO[M8 + i] = mx;
        }
        for (int i = tid; i < APQ_WAVES * 8 * APQ_PS; i += nt)
            reinterpret_cast<float *>(ap_smem + P.off_plane)[i] = 0.0f;         // incl. the zero tails of the planes
    }
    // per-lane constants
    ap_float2 win[25];
#pragma unroll
    for (int r = 0; r < 25; ++r) win[r] = reinterpret_cast<const ap_float2 *>(P.window)[q + 8 * r];
    const int qa = q & 3;
    // lane ^ 4 stage on the folded values: new = own + partner * kap, kap = W_8^qa for the difference
    // half (q >= 4: the partner holds sg u), -conj W_8^qa for the sum half (the partner holds -sg W u)
    const ap_float2 w8 = P.tw[50 * qa];
    const ap_float2 kap = q < 4 ? ap_mk(-w8.x, w8.y) : w8;
    const float s1 = qa < 2 ? 1.0f : -1.0f, s2 = (qa & 1) ? -1.0f : 1.0f;
    const float c1 = -s1, c2 = -s2;                            // stage coefficients on values held as (s1 s2) v
    const ap_float2 rotw = qa == 3 ? ap_mk(0.0f, 1.0f) : ap_mk(1.0f, 0.0f);
    const int k2 = 2 * (((q & 1) << 1) | ((q >> 1) & 1)) + (q >> 2);            // block of bins this lane ends up with
    // partner of register 0 (bins 25 k2 <-> 25 (8 - k2)): the lane of this frame whose block is (8 - k2) % 8
    const int k2p = (8 - k2) & 7;
    const int qp = (((k2p >> 1) & 1) << 1 | ((k2p >> 2) & 1)) | ((k2p & 1) << 2);
    const int src0 = (lane & ~7) | qp;
    const ap_float2 half = ap_mk(0.5f, 0.5f);
    AP_LDS_BARRIER();

    const int64_t worker = (int64_t)blockIdx.x * APQ_WAVES + wave;
    const int64_t n_workers = (int64_t)gridDim.x * APQ_WAVES;
    const int64_t grp_lo = P.n_groups * worker / n_workers, grp_hi = P.n_groups * (worker + 1) / n_workers;
    float vmax = -INFINITY;
    const int Ti = (int)P.T;
    if (grp_lo < grp_hi) {
        int64_t b = grp_lo / P.groups_per_clip;
        int t0 = (int)(grp_lo - b * P.groups_per_clip) * 8;
        ApClip clip = ap_clip_make(P.y + b * P.L, P.L);
        ap_float2 raw[25];
        auto load_group = [&](int tt0) {
            const int base = (tt0 + g) * P.hop - P.pad + 2 * q;
#pragma unroll
            for (int r = 0; r < 25; ++r) raw[r] = ap_clip_load2(clip, base + 16 * r);
        };
        load_group(t0);
        for (int64_t grp = grp_lo; grp < grp_hi; ++grp) {
            ap_float2 v[25];
#pragma unroll
            for (int r = 0; r < 25; ++r) v[r] = ap_mul2(raw[r], win[r]);
            const bool clip_ends = t0 + 8 >= Ti;
            AP_SCHED_FENCE();
            if (grp + 1 < grp_hi) {                           // next group, in flight during this one
                if (clip_ends) {
                    clip = ap_clip_make(P.y + (b + 1) * P.L, P.L);
                    load_group(0);
                } else {
                    load_group(t0 + 8);
                }
            }
            AP_SCHED_FENCE();
            apq_dft25(v);
            {
                ap_float2 t[25];
#pragma unroll
                for (int k1 = 0; k1 < 25; ++k1) t[k1] = T200[q * 25 + k1];
#pragma unroll
                for (int k1 = 0; k1 < 25; ++k1) v[k1] = ap_mul_fw(v[k1], t[k1]);
            }
            // ---- radix-8 across the 8 lanes of the frame ------------------------------------
#pragma unroll
            for (int k1 = 0; k1 < 25; ++k1) {                 // lane ^ 4: sums / twiddled differences
                const ap_float2 p = ap_mk(apq_xor4(v[k1].x), apq_xor4(v[k1].y));
                v[k1] = ap_cmul_tail_fw(p, kap, ap_fma_s(p, kap.x, v[k1]));
            }
            // quad radix-4: r = h - s1 h[lane ^ 2]; lane 3: r *= -i; out = r - s2 r[lane ^ 1]
            {
                // quad radix-4 on DPP moves + packed FMAs, 5 values at a time (values are held as (s1 s2) v).
                // (The v_fmac_f32_dpp form used by the 2048 kernel measured 15 % slower here: the asm blocks pin
                // 26 registers each and the allocator spills.)
#pragma unroll
                for (int h = 0; h < 25; h += 5) {
                    ap_float2 p[5];
                    AP_SCHED_FENCE();
#pragma unroll
                    for (int i = 0; i < 5; ++i) p[i] = ap_mk(ap_quad_xor2(v[h + i].x), ap_quad_xor2(v[h + i].y));
#pragma unroll
                    for (int i = 0; i < 5; ++i) v[h + i] = ap_fma_s(p[i], c1, v[h + i]);
                    AP_SCHED_FENCE();
#pragma unroll
                    for (int i = 0; i < 5; ++i) p[i] = ap_scale(v[h + i], rotw.x);
#pragma unroll
                    for (int i = 0; i < 5; ++i) v[h + i] = ap_cmul_tail_fw(v[h + i], rotw, p[i]);
                    AP_SCHED_FENCE();
#pragma unroll
                    for (int i = 0; i < 5; ++i) p[i] = ap_mk(ap_quad_xor1(v[h + i].x), ap_quad_xor1(v[h + i].y));
#pragma unroll
                    for (int i = 0; i < 5; ++i) v[h + i] = ap_fma_s(p[i], c2, v[h + i]);
                    AP_SCHED_FENCE();
                }
            }
            // ---- paired real split + power: this lane's bins k = k1 + 25 k2 -----------------
            //   X[k] = (Z[k] + conj Z[200-k]) / 2 + (-i) (W_400^k / 2) (Z[k] - conj Z[200-k])
            float *pl = plane + g * APQ_PS + 25 * k2;
            {
                const ap_float2 zm0 = ap_mk(apq_lane_read(v[0].x, src0), apq_lane_read(v[0].y, src0));
                ap_float2 sw[25];
#pragma unroll
                for (int k1 = 0; k1 < 25; ++k1) sw[k1] = S400[q * 25 + k1];
#pragma unroll
                for (int k1 = 0; k1 < 25; ++k1) {
                    const ap_float2 zk = v[k1];
                    const ap_float2 zm = k1 == 0 ? zm0 : ap_mk(apq_mirror8(v[25 - k1].x), apq_mirror8(v[25 - k1].y));
                    const ap_float2 a = ap_add_conj(zk, zm), d = ap_sub_conj(zk, zm);
                    const ap_float2 x = ap_fma_add_mi(a, half, ap_mul_fw(d, sw[k1]));
                    pl[k1] = apw_pow2x<PMODE>(x.x, x.y, P.power);
                }
                if (q == 0) {                                  // bin 200: X[200] = Re Z[0] - Im Z[0]
                    const float n = v[0].x - v[0].y;
                    plane[g * APQ_PS + APQ_NC] = apw_pow2x<PMODE>(n, 0.0f, P.power);
                }
            }
            AP_WAVE_SYNC();
            // ---- mel contraction: frame g, filters m = q + 8 i --------------------------------
            const int t = t0 + g;
            const float *prow = plane + g * APQ_PS;
            float *ob = P.out + b * (int64_t)M * P.T + t;
            // Four steps (32 filters) at a time: one trip of the chunk loop issues 4 weight quads + 16
            // plane values before it uses any of them, so the wave pays one LDS round trip per 4 bins of
            // the longest band in the set instead of one per 4 bins of every filter.
            for (int i0 = 0; i0 < NI; i0 += 4) {
                const float *pp[4], *wr[4];
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                int nmax = 0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int i = i0 + s < NI ? i0 + s : NI - 1;          // a clamped step repeats the last one (not stored)
                    const int m = 8 * i + q;
                    const int n = AP_UNIFORM(LO[M8 + i]);
                    nmax = n > nmax ? n : nmax;
                    pp[s] = prow + LO[m];
                    wr[s] = WT + m * APQ_WMAX;
                }
                for (int j = 0; j < nmax; j += 4) {
                    ap_float4 w[4];
                    float pv[4][4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        w[s] = *reinterpret_cast<const ap_float4 *>(wr[s] + j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[s][e] = pp[s][j + e];
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[s] = fmaf(w[s].x, pv[s][0], acc[s]);
                        acc[s] = fmaf(w[s].y, pv[s][1], acc[s]);
                        acc[s] = fmaf(w[s].z, pv[s][2], acc[s]);
                        acc[s] = fmaf(w[s].w, pv[s][3], acc[s]);
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int m = 8 * (i0 + s) + q;
                    if (i0 + s < NI && m < M && t < Ti) {
                        ob[(int64_t)m * P.T] = acc[s];
                        vmax = fmaxf(vmax, acc[s]);
                    }
                }
            }
            AP_WAVE_SYNC();
            if (clip_ends) { t0 = 0; ++b; } else { t0 += 8; }
        }
    }
    if (P.max_key) {                  // one atomic per wave: lanes -> LDS -> lane 0
        AP_WAVE_SYNC();
        plane[lane] = vmax;
        AP_WAVE_SYNC();
        if (lane == 0) {
            float m = plane[0];
            for (int i = 1; i < 64; ++i) m = fmaxf(m, plane[i]);
            ap_atomic_max_u32(P.max_key, ap_fkey(m));
        }
    }
}
