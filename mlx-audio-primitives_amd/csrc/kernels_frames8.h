// Eight frames per wave: fused mel-spectrogram and STFT kernels for n_fft = 16 R on gfx950, R = 25
// (n_fft 400, the Whisper front end: hop 160, 80 filters), R = 32 (n_fft 512) and R = 16 (n_fft 256).
// One wave64 transforms EIGHT consecutive frames at a time, 8 lanes per frame, R complex values per
// lane, with no workgroup barrier and no LDS pass inside the transform.
//
// The packed (8 R)-point complex transform z[m] = x[2m] + i x[2m+1] is R x 8:
//
//   lane (g = lane >> 3, q = lane & 7) holds z_g[q + 8 r], r = 0..R-1  (bounds-checked buffer loads:
//   the hardware range check is the constant padding)
//   radix-R in registers (5 x 5, 4 x 8 or 4 x 4, compile-time twiddles)   [mx.fft.rfft, stft.py:130]
//   * W_8R^(q k1) (small LDS table)
//   radix-8 ACROSS the 8 lanes of a frame: one ds_swizzle (lane ^ 4) stage, then the quad radix-4
//   with DPP quad_perm as in kernels_wave.h -> lane q owns the R contiguous bins
//   k = k1 + R k2(q), k2(q) = 2 bitrev2(q & 3) + (q >> 2)
//   paired real split: Z[8R - k] sits in lane 7 - q (DPP row_half_mirror), register R - k1
//   STFT: X[k] goes straight to HBM (8 lanes g = 8 consecutive frames = one 64-byte run per bin)
//   mel: |X|^p -> the wave's 8 power planes in LDS, then lane (g, q) owns frame g and the filters
//   m = q + 8 i: a dot product over the filter's band with zero-padded weights from an LDS table (the
//   8 lanes of a step hold 8 adjacent filters, so their band lengths agree to within a few bins),
//   four steps per trip so that the LDS reads of 32 filters are in flight together.
//
// LDS layout of a plane (and of the table rows): for even R the blocks of R bins are one float apart
// (block stride R + 1) - with stride 32 the 8 lanes of a frame would write one bank.  The band table
// is built in that padded address space (weight 0 on the pad slots), so band reads stay contiguous.
//
// n_fft = 400: ~160 wave-instructions per frame against ~360 (and four workgroup barriers per 8
// frames) in the compile-time LDS engine that served these shapes before (kernels_ct.h).  Constant
// padding / center=False, n_mels <= 128, band lengths <= 64.  Reference: mel.py:245-352, stft.py:92-135.
#pragma once
#include "kernels_wave.h"

#define APQ_WAVES 8          // waves per workgroup (they only share read-only LDS tables)

template <int R>
struct ApqGeom {
    static constexpr int NC = 8 * R;                         // complex points
    static constexpr int PADB = (R % 2 == 0) ? 1 : 0;        // pad floats after each block of R bins
    static constexpr int BS = R + PADB;                      // block stride (planes and table rows)
    static constexpr int WMAX = 64 + 4 * PADB;               // most floats per filter row of the LDS weight table
    static constexpr int NEED = 8 * BS + 1 + WMAX;           // bins + Nyquist + zero tail for padded band reads
    static constexpr int PS = ((NEED - 8 + 31) / 32) * 32 + 8;   // floats per power plane, = 8 (mod 32)
    static constexpr bool WIN_REGS = R <= 25;                // window pairs in registers (else an LDS table)
    static constexpr int A = R == 25 ? 5 : 4;                // radix-R = A x BN
    static constexpr int BN = R / A;
};

struct ApFrames8Params {
    const float *y;            // (B, L)
    const float *window;       // (n_fft)
    const ap_float2 *tw;       // (n_fft) (cos, sin)(2 pi j / n_fft)
    const float *fb;           // (M, n_fft/2 + 1) dense filterbank          (mel)
    const int32_t *band_lo, *band_len;   // (M) span of each filter's non-zeros (mel)
    float *out;                // mel: (B, M, T); STFT: (B, n_fft/2 + 1, T) complex64
    unsigned *max_key;
    int64_t L, T, n_clips, groups_per_clip, n_groups;
    int64_t Ts;                // elements between the rows of `out` (T = dense): complex values for the STFT (a multiple of 16 =
                               // whole lines), floats for mel (a multiple of 8 = the 8-frame runs are whole sectors)
    int plain_stores;          // mel: 1 = no lane transpose before the stores (default; AP_MEL8_TRANSPOSED_STORES=1 turns it on)
    int hop, pad, pad_mode, n_mels, wmax;   // pad_mode: used by the PADGEN instantiations only.   wmax: floats per filter row of the LDS weight table (32 or 64, + 4 for even R)
    float power;
    int off_t, off_s, off_win, off_w, off_lo, off_plane, off_stage, lds_bytes;   // off_stage: 0 = no output tile
};

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

#define APQ_CR(e, R) ((float)__builtin_cos(6.283185307179586476925 * (e) / (double)(R)))
#define APQ_SR(e, R) ((float)__builtin_sin(6.283185307179586476925 * (e) / (double)(R)))

// R-point DFT in registers: r = r0 + A r1 in, k = kappa + BN k0 out
template <int R>
AP_DEV void apq_dft(ap_float2 (&v)[R]) {
    constexpr int A = ApqGeom<R>::A, BN = ApqGeom<R>::BN;
    ap_float2 a[A][BN];
#pragma unroll
    for (int r0 = 0; r0 < A; ++r0) {
        ap_float2 t[BN];
#pragma unroll
        for (int r1 = 0; r1 < BN; ++r1) t[r1] = v[r0 + A * r1];
        ApButterfly<BN>::run(t);
#pragma unroll
        for (int k = 0; k < BN; ++k)
            a[r0][k] = (r0 > 0 && k > 0) ? ap_mul_fw_c(t[k], APQ_CR(r0 * k, R), APQ_SR(r0 * k, R)) : t[k];
    }
#pragma unroll
    for (int k = 0; k < BN; ++k) {
        ap_float2 u[A];
#pragma unroll
        for (int r0 = 0; r0 < A; ++r0) u[r0] = a[r0][k];
        ApButterfly<A>::run(u);
#pragma unroll
        for (int k0 = 0; k0 < A; ++k0) v[k + BN * k0] = u[k0];
    }
}

// per-lane constants of the radix-8 across the 8 lanes of a frame
struct ApqLane {
    int g, q, k2, src0;
    ap_float2 kap, rotw;
    float c1, c2;
};
AP_DEV int apq_block_of(int q) { return 2 * (((q & 1) << 1) | ((q >> 1) & 1)) + (q >> 2); }
AP_DEV ApqLane apq_lane_make(int lane, const ap_float2 *tw, int R) {
    ApqLane Ln;
    Ln.g = lane >> 3;
    Ln.q = lane & 7;
    const int qa = Ln.q & 3;
    // lane ^ 4 stage on the folded values: new = own + partner * kap, kap = W_8^qa for the difference
    // half (q >= 4: the partner holds sg u), -conj W_8^qa for the sum half (the partner holds -sg W u)
    const ap_float2 w8 = tw[2 * R * qa];
    Ln.kap = Ln.q < 4 ? ap_mk(-w8.x, w8.y) : w8;
    const float s1 = qa < 2 ? 1.0f : -1.0f, s2 = (qa & 1) ? -1.0f : 1.0f;
    Ln.c1 = -s1;                                              // stage coefficients on values held as (s1 s2) v
    Ln.c2 = -s2;
    Ln.rotw = qa == 3 ? ap_mk(0.0f, 1.0f) : ap_mk(1.0f, 0.0f);
    Ln.k2 = apq_block_of(Ln.q);                               // block of bins this lane ends up with
    // partner of register 0 (bins R k2 <-> R (8 - k2)): the lane of this frame whose block is (8 - k2) % 8
    const int k2p = (8 - Ln.k2) & 7;
    const int qp = (((k2p >> 1) & 1) << 1 | ((k2p >> 2) & 1)) | ((k2p & 1) << 2);
    Ln.src0 = (lane & ~7) | qp;
    return Ln;
}

// the two twiddle tables (rows of stride BS): T[q][k1] = W_8R^(q k1) times the lane's own factor of the
// lane ^ 4 stage (sg = s1 s2 of the quad stage: sg for q < 4, the lane keeps sg u_q; -sg W_8^(q & 3) for
// q >= 4, it keeps -sg W u_q);  S[q][k1] = W_16R^(k1 + R k2(q)) / 2
template <int R>
AP_DEV void apq_fill_tables(ap_float2 *Tt, ap_float2 *St, const ap_float2 *tw, int tid, int nt) {
    constexpr int BS = ApqGeom<R>::BS;
    for (int i = tid; i < 8 * R; i += nt) {
        const int qq = i / R, k1 = i - qq * R;
        const int ja = qq & 3;
        const float sg = (ja == 1 || ja == 2) ? -1.0f : 1.0f;
        const ap_float2 t = tw[(2 * qq * k1) % (16 * R)];
        const ap_float2 f = qq < 4 ? ap_mk(sg, 0.0f) : ap_scale(tw[2 * R * ja], -sg);
        Tt[qq * BS + k1] = ap_mk(t.x * f.x - t.y * f.y, t.x * f.y + t.y * f.x);     // product of two forward-form twiddles
        St[qq * BS + k1] = ap_scale(tw[k1 + R * apq_block_of(qq)], 0.5f);
    }
}

// windowed samples -> this lane's R bins Z[k1 + R k2] of the packed transform (values carry the factor
// the table folded in: see apq_fill_tables)
template <int R>
AP_DEV void apq_transform(ap_float2 (&v)[R], const ap_float2 *Trow, const ApqLane &Ln) {
    apq_dft<R>(v);
    {
        ap_float2 t[R];
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) t[k1] = Trow[k1];
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) v[k1] = ap_mul_fw(v[k1], t[k1]);
    }
    // ---- radix-8 across the 8 lanes of the frame ------------------------------------
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) {                 // lane ^ 4: sums / twiddled differences
        const ap_float2 p = ap_mk(apq_xor4(v[k1].x), apq_xor4(v[k1].y));
        v[k1] = ap_cmul_tail_fw(p, Ln.kap, ap_fma_s(p, Ln.kap.x, v[k1]));
    }
    // quad radix-4: r = h - s1 h[lane ^ 2]; lane 3: r *= -i; out = r - s2 r[lane ^ 1], on DPP moves +
    // packed FMAs, a few values at a time (values are held as (s1 s2) v).  (The v_fmac_f32_dpp form used
    // by the 2048 kernel measured 15 % slower here: the asm blocks pin 26 registers each and the
    // allocator spills.)
    constexpr int H = R % 5 == 0 ? 5 : 4;
#pragma unroll
    for (int h = 0; h < R; h += H) {
        ap_float2 p[H];
        AP_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < H; ++i) p[i] = ap_mk(ap_quad_xor2(v[h + i].x), ap_quad_xor2(v[h + i].y));
#pragma unroll
        for (int i = 0; i < H; ++i) v[h + i] = ap_fma_s(p[i], Ln.c1, v[h + i]);
        AP_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < H; ++i) p[i] = ap_scale(v[h + i], Ln.rotw.x);
#pragma unroll
        for (int i = 0; i < H; ++i) v[h + i] = ap_cmul_tail_fw(v[h + i], Ln.rotw, p[i]);
        AP_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < H; ++i) p[i] = ap_mk(ap_quad_xor1(v[h + i].x), ap_quad_xor1(v[h + i].y));
#pragma unroll
        for (int i = 0; i < H; ++i) v[h + i] = ap_fma_s(p[i], Ln.c2, v[h + i]);
        AP_SCHED_FENCE();
    }
}

// paired real split of the lane's bins k = k1 + R k2:
//   X[k] = (Z[k] + conj Z[8R-k]) / 2 + (-i) (W_16R^k / 2) (Z[k] - conj Z[8R-k]);  emit(k1, X[k])
template <int R, class Emit>
AP_DEV void apq_split(const ap_float2 (&v)[R], const ap_float2 *Srow, const ApqLane &Ln, Emit &&emit) {
    const ap_float2 half = ap_mk(0.5f, 0.5f);
    const ap_float2 zm0 = ap_mk(apq_lane_read(v[0].x, Ln.src0), apq_lane_read(v[0].y, Ln.src0));
    ap_float2 sw[R];
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) sw[k1] = Srow[k1];
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) {
        const ap_float2 zk = v[k1];
        const ap_float2 zm = k1 == 0 ? zm0 : ap_mk(apq_mirror8(v[R - k1].x), apq_mirror8(v[R - k1].y));
        const ap_float2 a = ap_add_conj(zk, zm), d = ap_sub_conj(zk, zm);
        emit(k1, ap_fma_add_mi(a, half, ap_mul_fw(d, sw[k1])));
    }
}

// one wave's stretch of 8-frame groups: sample loads one group ahead, window, transform; body(v, b, t0)
// gets the lane's bins of the packed transform
// PADGEN: reflect / edge padding and odd hops - a lane whose frame reaches over a clip end loads through the
// index remap (ap_load_padded), the others through the bounds-checked loads
template <int R, int PADGEN, class Body>
AP_DEV void apq_group_loop(const ApFrames8Params &P, const ApqLane &Ln, const ap_float2 *Trow,
                           const ap_float2 *WINP, int wave, Body &&body) {
    constexpr bool WIN_REGS = ApqGeom<R>::WIN_REGS;
    ap_float2 win[R];                                          // (dead when the window lives in LDS)
    if (WIN_REGS) {
#pragma unroll
        for (int r = 0; r < R; ++r) win[r] = reinterpret_cast<const ap_float2 *>(P.window)[Ln.q + 8 * r];
    }
    const int64_t worker = (int64_t)blockIdx.x * APQ_WAVES + wave;
    const int64_t n_workers = (int64_t)gridDim.x * APQ_WAVES;
    const int64_t grp_lo = P.n_groups * worker / n_workers, grp_hi = P.n_groups * (worker + 1) / n_workers;
    const int Ti = (int)P.T;
    if (grp_lo >= grp_hi) return;
    int64_t b = grp_lo / P.groups_per_clip;
    int t0 = (int)(grp_lo - b * P.groups_per_clip) * 8;
    ApClip clip = ap_clip_make(P.y + b * P.L, P.L);
    ap_float2 raw[R];
    auto load_group = [&](int64_t bb, int tt0) {
        const int fbase = (tt0 + Ln.g) * P.hop - P.pad;
        const int base = fbase + 2 * Ln.q;
        if (PADGEN && !(fbase >= 0 && (int64_t)fbase + 16 * R <= P.L)) {
            const float *yb = P.y + bb * P.L;
#pragma unroll
            for (int r = 0; r < R; ++r)
                raw[r] = ap_mk(ap_load_padded(yb, P.L, base + 16 * r, P.pad_mode), ap_load_padded(yb, P.L, base + 16 * r + 1, P.pad_mode));
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) raw[r] = ap_clip_load2(clip, base + 16 * r);
        }
    };
    load_group(b, t0);
    // (the loop leaves through `more`: were the prefetch skipped on a path that re-enters the loop, the
    // register allocator would copy all R pairs at that merge)
    for (int64_t grp = grp_lo;; ++grp) {
        ap_float2 v[R];
        if (WIN_REGS) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = ap_mul2(raw[r], win[r]);
        } else {
            ap_float2 w[R];
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = WINP[Ln.q + 8 * r];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = ap_mul2(raw[r], w[r]);
        }
        const bool clip_ends = t0 + 8 >= Ti;
        // the windowed values exist before the prefetch is issued, so the samples' registers are free for
        // it (otherwise the multiplies sink below the loads and the prefetch needs a second register set
        // plus R copies per group)
#pragma unroll
        for (int r = 0; r < R; ++r) AP_PIN(v[r]);
        AP_SCHED_FENCE();
        const bool more = grp + 1 < grp_hi;
        if (more) {                                       // next group, in flight during this one
            if (clip_ends) clip = ap_clip_make(P.y + (b + 1) * P.L, P.L);
            load_group(clip_ends ? b + 1 : b, clip_ends ? 0 : t0 + 8);
        }
        AP_SCHED_FENCE();
        apq_transform<R>(v, Trow, Ln);
        body(v, b, t0);
        if (!more) break;
        if (clip_ends) { t0 = 0; ++b; } else { t0 += 8; }
    }
}

// ---------------------------------------------------------------------------------------------
// mel-spectrogram
// ---------------------------------------------------------------------------------------------
template <int R, int PMODE, int PADGEN = 0>
__global__ void __launch_bounds__(64 * APQ_WAVES, 2) ap_mel8_wave_kernel(ApFrames8Params P) {
    typedef ApqGeom<R> G;
    constexpr int NC = G::NC, BS = G::BS, PS = G::PS, PADB = G::PADB;
    const int WMAX = P.wmax;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *Tt = reinterpret_cast<ap_float2 *>(ap_smem + P.off_t);           // [8][BS]
    ap_float2 *St = reinterpret_cast<ap_float2 *>(ap_smem + P.off_s);           // [8][BS]
    ap_float2 *WINP = reinterpret_cast<ap_float2 *>(ap_smem + P.off_win);       // [8 R] window pairs (R > 25)
    float *WT = reinterpret_cast<float *>(ap_smem + P.off_w);                   // [M8][WMAX] zero-padded band weights
    int *LO = reinterpret_cast<int *>(ap_smem + P.off_lo);                      // [M8] band start; [M8 + i]: longest band of step i
    float *plane = reinterpret_cast<float *>(ap_smem + P.off_plane) + wave * (8 * PS);
    const int M = P.n_mels;
    const int NI = (M + 7) / 8;                 // contraction steps: filters 8 i + q
    const int M8 = 8 * NI;
    // plane address of bin k: k + PADB (k / R)
    auto padded = [](int k) { return k + PADB * (k / R); };

    // ---------------- workgroup tables (once; the only workgroup barrier) ----------------
    {
        const int nt = 64 * APQ_WAVES;
        apq_fill_tables<R>(Tt, St, P.tw, tid, nt);
        if (!G::WIN_REGS)
            for (int i = tid; i < NC; i += nt) WINP[i] = reinterpret_cast<const ap_float2 *>(P.window)[i];
        for (int i = tid; i < M8 * WMAX; i += nt) {
            const int m = i / WMAX, j = i - m * WMAX;
            float w = 0.0f;
            if (m < M) {
                const int lo = P.band_lo[m], len = P.band_len[m];
                const int a = padded(lo) + j;                       // plane address of this weight
                const int blk = a / BS < 8 ? a / BS : 8;
                const int k = a - PADB * blk;
                const bool pad_slot = PADB && blk < 8 && a - blk * BS == R;
                if (!pad_slot && k >= lo && k < lo + len) w = P.fb[(int64_t)m * (NC + 1) + k];
            }
            WT[i] = w;
        }
        for (int i = tid; i < M8; i += nt) LO[i] = i < M ? padded(P.band_lo[i]) : 0;
        for (int i = tid; i < NI; i += nt) {
            int mx = 0;
            for (int j = 0; j < 8; ++j) {
                const int m = 8 * i + j;
                if (m < M && P.band_len[m] > 0) {
                    const int lo = P.band_lo[m], n = padded(lo + P.band_len[m] - 1) - padded(lo) + 1;
                    if (n > mx) mx = n;
                }
            }
            LO[M8 + i] = mx < WMAX ? mx : WMAX;
        }
        for (int i = tid; i < APQ_WAVES * 8 * PS; i += nt)
            reinterpret_cast<float *>(ap_smem + P.off_plane)[i] = 0.0f;         // incl. pad slots and the zero tails
    }
    const ApqLane Ln = apq_lane_make(lane, P.tw, R);
    const int g = Ln.g, q = Ln.q;
    AP_LDS_BARRIER();

    float vmax = -INFINITY;
    const int Ti = (int)P.T;
    float *stage = P.off_stage ? reinterpret_cast<float *>(ap_smem + P.off_stage) + wave * (M8 * 8) : nullptr;
    const int gt = lane & 7, qt = lane >> 3;                  // store role: frame gt of the rows 8 i + qt
    const int lane_t = (gt << 3) | qt;                        // ... whose value lane (g = gt, q = qt) computed
    apq_group_loop<R, PADGEN>(P, Ln, Tt + q * BS, WINP, wave, [&](ap_float2 (&v)[R], int64_t b, int t0) {
        // ---- paired real split + power: this lane's bins k = k1 + R k2 --------------------
        float *pl = plane + g * PS + BS * Ln.k2;
        apq_split<R>(v, St + q * BS, Ln, [&](int k1, ap_float2 x) { pl[k1] = apw_pow2x<PMODE>(x.x, x.y, P.power); });
        if (q == 0) {                                  // bin 8R: X = Re Z[0] - Im Z[0]
            const float n = v[0].x - v[0].y;
            plane[g * PS + 8 * BS] = apw_pow2x<PMODE>(n, 0.0f, P.power);
        }
        AP_WAVE_SYNC();
        // ---- mel contraction: frame g, filters m = q + 8 i --------------------------------
        const int t = t0 + g;
        const float *prow = plane + g * PS;
        float *obt = P.out + b * (int64_t)M * P.Ts + t0 + gt;   // rows Ts floats apart (T = dense); transposed store role
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
                wr[s] = WT + m * WMAX;
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
            // Lane (g, q) = 8 g + q has frame g of row 8 i + q: stored as they are, the eight lanes of a row's 32-byte
            // run are eight lanes apart and the write path does not merge them (WRITE_SIZE 1.6-1.8 x the array).  One
            // lane permute per value turns the 8 x 8 lane block over: lane 8 q + g stores (row 8 i + q, frame g), so a
            // run is eight ADJACENT lanes = one request.
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = 8 * (i0 + s) + q;
                if (i0 + s < NI && m < M && t < Ti) vmax = fmaxf(vmax, acc[s]);
                if (stage) {                                       // (uniform) into the wave's output tile
                    if (i0 + s < NI) stage[m * 8 + g] = acc[s];
                    continue;
                }
                if (P.plain_stores) {                              // (uniform): every lane stores its own value
                    if (i0 + s < NI && m < M && t < Ti) P.out[b * (int64_t)M * P.Ts + t + (int64_t)m * P.Ts] = acc[s];
                    continue;
                }
                const float v = apq_lane_read(acc[s], lane_t);
                const int mt = 8 * (i0 + s) + qt;
                if (i0 + s < NI && mt < M && t0 + gt < Ti) obt[(int64_t)mt * P.Ts] = v;
            }
        }
        AP_WAVE_SYNC();
        if (stage) {
            // the tile leaves as 16-byte pieces: lane pair (2 r, 2 r + 1) stores the two halves of row r's 8-frame run
            // (a quarter of the store instructions, and the write path sees 32 contiguous bytes from adjacent lanes)
            float *orow = P.out + b * (int64_t)M * P.Ts + t0;
            for (int e = lane; e < 2 * M; e += 64) {
                const int row = e >> 1, h4 = (e & 1) * 4;
                const ap_float4 v4 = *reinterpret_cast<const ap_float4 *>(stage + row * 8 + h4);
                float *dst = orow + (int64_t)row * P.Ts + h4;
                if (t0 + h4 + 3 < Ti) {
                    ap_rsp_f4u u;
                    u.x = v4.x; u.y = v4.y; u.z = v4.z; u.w = v4.w;
                    *reinterpret_cast<ap_rsp_f4u *>(dst) = u;
                } else {
                    if (t0 + h4 < Ti) dst[0] = v4.x;
                    if (t0 + h4 + 1 < Ti) dst[1] = v4.y;
                    if (t0 + h4 + 2 < Ti) dst[2] = v4.z;
                }
            }
            AP_WAVE_SYNC();
        }
    });
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

// ---------------------------------------------------------------------------------------------
// STFT: out (B, 8R + 1, T) complex64
// ---------------------------------------------------------------------------------------------
template <int R, int PADGEN = 0>
__global__ void __launch_bounds__(64 * APQ_WAVES, 2) ap_stft8_wave_kernel(ApFrames8Params P) {
    typedef ApqGeom<R> G;
    constexpr int NC = G::NC, BS = G::BS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *Tt = reinterpret_cast<ap_float2 *>(ap_smem + P.off_t);
    ap_float2 *St = reinterpret_cast<ap_float2 *>(ap_smem + P.off_s);
    ap_float2 *WINP = reinterpret_cast<ap_float2 *>(ap_smem + P.off_win);
    {
        const int nt = 64 * APQ_WAVES;
        apq_fill_tables<R>(Tt, St, P.tw, tid, nt);
        if (!G::WIN_REGS)
            for (int i = tid; i < NC; i += nt) WINP[i] = reinterpret_cast<const ap_float2 *>(P.window)[i];
    }
    const ApqLane Ln = apq_lane_make(lane, P.tw, R);
    AP_LDS_BARRIER();
    const int Ti = (int)P.T;
    // One clip's output rows as a bounds-checked buffer (the launch code keeps them under 2 GiB): the
    // lane part of a store's address is (bin block R k2, frame g), the uniform part (bin k1, group t0).
    const int Tsi = (int)P.Ts;                                                  // row stride (T, or padded to whole lines)
    const unsigned lane_bytes = 8u * ((unsigned)(R * Ln.k2) * (unsigned)Tsi + (unsigned)Ln.g);
    const int64_t clip_bytes = (int64_t)(NC + 1) * P.Ts * 8;
    apq_group_loop<R, PADGEN>(P, Ln, Tt + Ln.q * BS, WINP, wave, [&](ap_float2 (&v)[R], int64_t b, int t0) {
        const ApOutBuf ob = ap_outbuf_make(reinterpret_cast<char *>(P.out) + b * clip_bytes, clip_bytes);
        const unsigned lb = t0 + Ln.g < Ti ? lane_bytes : 0xF0000000u;          // frames past T: parked
        apq_split<R>(v, St + Ln.q * BS, Ln, [&](int k1, ap_float2 x) {
            ap_outbuf_store2(ob, lb, 8u * (unsigned)(k1 * Tsi + t0), x);
        });
        if (Ln.q == 0) ap_outbuf_store2(ob, lb, 8u * (unsigned)(NC * Tsi + t0), ap_mk(v[0].x - v[0].y, 0.0f));   // bin 8R
    });
}

// ---------------------------------------------------------------------------------------------
// Inverse: frames (B, T, 16 R) float32 from the spectrum (B, 8R + 1, T) complex64 (mx.fft.irfft, stft.py:295)
// ---------------------------------------------------------------------------------------------
// The forward machinery is reused as it stands: the spectrum is LOADED in the transform's input layout
// (lane (g, q), register r = bin q + 8 r of frame g), merged with its Hermitian mirror into the packed
// half-length spectrum - conjugated - and sent through apq_transform; lane q then holds
// conj(NC z[m]) for the R contiguous m = k1 + R k2(q), i.e. the 2 R contiguous samples
// x[2m] = Re / NC, x[2m + 1] = -Im / NC of the frame.
//   conj Z[k] = (Xm + conj X) / 2 - (-i) (W_16R^k / 2) (Xm - conj X),  Xm = X[8R - k]
// (the mirror of bin q + 8 r is bin (8 - q) + 8 (R - 1 - r): lane 8 - q of the frame, fetched with one
// ds_bpermute per component; lane q = 0 mirrors onto itself, register R - r, and onto the Nyquist bin for
// r = 0; the imaginary parts of bins 0 and 8R are ignored like numpy / mlx irfft do).
struct ApIrfft8Params {
    const ap_float2 *S;        // (B, 8R + 1, T)
    const ap_float2 *tw;       // (16 R)
    float *frames;             // (B, T, 16 R)
    int64_t T, n_clips, groups_per_clip, n_groups;
    int off_t, off_s, lds_bytes;
};

// spectrum rows of frame t0 + g (lane part lb of the address; lanes of frames past T hold an out-of-range
// one and read zeros) -> conj(NC z[m]) of the lane's R contiguous m
template <int R>
AP_DEV void apq_inverse_frame(ApOutBuf sb, unsigned lb, int Ti, int t0, const ap_float2 *Smrow, const ap_float2 *Trow,
                              const ApqLane &Ln, int msrc, ap_float2 (&v)[R]) {
    const int q = Ln.q;
    const ap_float2 half = ap_mk(0.5f, 0.5f);
    ap_float2 X[R + 1];
#pragma unroll
    for (int r = 0; r < R; ++r) X[r] = ap_outbuf_load2(sb, lb, 8u * (unsigned)(8 * r * Ti + t0));
    // row 8R, the Nyquist bin: lane q = 0 only (the uniform part of an address is not range-checked, so the
    // other lanes must not reach past their clip's rows)
    X[R] = ap_outbuf_load2(sb, q == 0 ? lb : 0xF0000000u, 8u * (unsigned)(8 * R * Ti + t0));
    if (q == 0) { X[0].y = 0.0f; X[R].y = 0.0f; }
    ap_float2 sw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) sw[r] = Smrow[r];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // what this lane offers to its mirror lane: register R - 1 - r (q = 0: its own register R - r)
        // (both candidates are pinned as values first: left alone, the compiler turns the select of two array
        // elements into one dynamically indexed element and the whole array moves to scratch memory)
        ap_float2 g0 = X[R - r], g1 = X[R - 1 - r];
        AP_PIN(g0);
        AP_PIN(g1);
        const ap_float2 give = q == 0 ? g0 : g1;
        const ap_float2 xm = ap_mk(apq_lane_read(give.x, msrc), apq_lane_read(give.y, msrc));
        const ap_float2 a = ap_add_conj(xm, X[r]), d = ap_sub_conj(xm, X[r]);
        v[r] = ap_fma_sub_mi(a, half, ap_mul_fw(d, sw[r]));
    }
    apq_transform<R>(v, Trow, Ln);
}

// the two LDS tables of the inverse kernels: Tt as in the forward kernels, Sm[q][r] = W_16R^(q + 8 r) / 2
template <int R>
AP_DEV void apq_fill_inverse_tables(ap_float2 *Tt, ap_float2 *Sm, const ap_float2 *tw, int tid, int nt) {
    constexpr int BS = ApqGeom<R>::BS;
    apq_fill_tables<R>(Tt, Sm, tw, tid, nt);                 // (Sm is overwritten just below)
    AP_LDS_BARRIER();
    for (int i = tid; i < 8 * R; i += nt) {
        const int qq = i / R, r = i - qq * R;
        Sm[qq * BS + r] = ap_scale(tw[qq + 8 * r], 0.5f);
    }
}

template <int R>
__global__ void __launch_bounds__(64 * APQ_WAVES, 2) ap_irfft8_wave_kernel(ApIrfft8Params P) {
    typedef ApqGeom<R> G;
    constexpr int NC = G::NC, BS = G::BS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *Tt = reinterpret_cast<ap_float2 *>(ap_smem + P.off_t);           // [8][BS]
    ap_float2 *Sm = reinterpret_cast<ap_float2 *>(ap_smem + P.off_s);           // [8][BS]
    apq_fill_inverse_tables<R>(Tt, Sm, P.tw, tid, 64 * APQ_WAVES);
    const ApqLane Ln = apq_lane_make(lane, P.tw, R);
    const int g = Ln.g, q = Ln.q;
    AP_LDS_BARRIER();
    const int Ti = (int)P.T;
    const int msrc = (lane & ~7) | ((8 - q) & 7);                               // lane holding the mirror bins
    const float sc = 1.0f / (float)NC;
    const ap_float2 scv = ap_mk(sc, -sc);                                       // conj and 1 / NC in one multiply
    const int64_t clip_bytes = (int64_t)(NC + 1) * P.T * 8;
    const unsigned lane_bytes = 8u * ((unsigned)q * (unsigned)Ti + (unsigned)g);

    const int64_t worker = (int64_t)blockIdx.x * APQ_WAVES + wave;
    const int64_t n_workers = (int64_t)gridDim.x * APQ_WAVES;
    const int64_t grp_lo = P.n_groups * worker / n_workers, grp_hi = P.n_groups * (worker + 1) / n_workers;
    for (int64_t grp = grp_lo; grp < grp_hi; ++grp) {
        const int64_t b = grp / P.groups_per_clip;
        const int t0 = (int)(grp - b * P.groups_per_clip) * 8;
        const bool live = t0 + g < Ti;
        const ApOutBuf sb = ap_outbuf_make(const_cast<char *>(reinterpret_cast<const char *>(P.S)) + b * clip_bytes, clip_bytes);
        ap_float2 v[R];
        apq_inverse_frame<R>(sb, live ? lane_bytes : 0xF0000000u, Ti, t0, Sm + q * BS, Tt + q * BS, Ln, msrc, v);
        if (live) {
            float *fr = P.frames + ((b * P.T + t0 + g) * (int64_t)(2 * NC)) + 2 * R * Ln.k2;
#pragma unroll
            for (int k1 = 0; k1 < R; ++k1) reinterpret_cast<ap_float2 *>(fr)[k1] = ap_mul2(v[k1], scv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ISTFT in one kernel (stft.py:292-338: irfft -> overlap-add -> window-sum-squares normalisation -> trim):
// the frames never reach HBM.  A wave walks a contiguous stretch of 8-frame groups; its windowed samples go
// into the wave's LDS accumulator (span 7 hop + n_fft samples from the group's first
// sample; the first n_fft - hop of them are what the previous group left), the 8 hop samples the group
// completes are normalised (window-sum-squares of the frames that exist: a table for interior positions, a
// sum over the <= n_fft / hop covering frames at the clip's ends) and stored with the centre trim folded
// in; the tail becomes the next group's carry.  The last group of a clip emits its tail as well and
// zero-fills what is left of the output row.  A stretch that starts inside a clip first runs the
// preceding group with its stores disabled to rebuild the carry (n_fft - hop <= 8 hop).
// ---------------------------------------------------------------------------------------------
struct ApIstft8Params {
    const ap_float2 *S;        // (B, 8R + 1, T)
    const ap_float2 *tw;       // (16 R)
    const float *window;       // (16 R)
    float *y;                  // (B, out_len)
    int64_t T, n_clips, groups_per_clip, n_groups, out_offset, out_len;
    int64_t Ts;                // complex values between the rows of S (T = dense)
    int hop, span;             // span = 7 hop + n_fft
    int off_t, off_s, off_w2, off_wss, off_acc, lds_bytes;
};


template <int R>
__global__ void __launch_bounds__(64 * APQ_WAVES, 2) ap_istft8_wave_kernel(ApIstft8Params P) {
    typedef ApqGeom<R> G;
    constexpr int NC = G::NC, BS = G::BS, N = 2 * NC;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *Tt = reinterpret_cast<ap_float2 *>(ap_smem + P.off_t);
    ap_float2 *Sm = reinterpret_cast<ap_float2 *>(ap_smem + P.off_s);
    float *W2 = reinterpret_cast<float *>(ap_smem + P.off_w2);                  // [N] window^2
    float *WSS = reinterpret_cast<float *>(ap_smem + P.off_wss);                // [hop] sum over all covering frames
    float *acc = reinterpret_cast<float *>(ap_smem + P.off_acc) + wave * P.span;
    const int hop = P.hop, span = P.span;
    {
        const int nt = 64 * APQ_WAVES;
        apq_fill_inverse_tables<R>(Tt, Sm, P.tw, tid, nt);
        for (int i = tid; i < N; i += nt) { const float w = P.window[i]; W2[i] = w * w; }
        for (int i = tid; i < hop; i += nt) {
            float s = 0.0f;
            // frames in increasing order = decreasing window index, like ap_overlap_add_kernel's loop
            int s0 = i;
            while (s0 + hop < N) s0 += hop;
            for (int j = s0; j >= 0; j -= hop) { const float w = P.window[j]; s += w * w; }
            WSS[i] = s;
        }
        for (int i = tid; i < APQ_WAVES * span; i += nt) reinterpret_cast<float *>(ap_smem + P.off_acc)[i] = 0.0f;
    }
    const ApqLane Ln = apq_lane_make(lane, P.tw, R);
    const int g = Ln.g, q = Ln.q;
    const int Ti = (int)P.T;
    const int msrc = (lane & ~7) | ((8 - q) & 7);
    const float sc = 1.0f / (float)NC;
    // the window pairs of this lane's samples with the 1 / NC and the conjugation folded in
    ap_float2 wv[R];
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) {
        const ap_float2 w = reinterpret_cast<const ap_float2 *>(P.window)[R * Ln.k2 + k1];
        wv[k1] = ap_mk(w.x * sc, -w.y * sc);
    }
    AP_LDS_BARRIER();
    const int Tsi = (int)P.Ts;                                                  // row stride of S
    const int64_t clip_bytes = (int64_t)(NC + 1) * P.Ts * 8;
    const unsigned lane_bytes = 8u * ((unsigned)q * (unsigned)Tsi + (unsigned)g);
    const int n_carry = N - hop;                                                // samples a group hands on
    const int n_round = (N + hop - 1) / hop < 8 ? (N + hop - 1) / hop : 8;

    const int64_t worker = (int64_t)blockIdx.x * APQ_WAVES + wave;
    const int64_t n_workers = (int64_t)gridDim.x * APQ_WAVES;
    const int64_t grp_lo = P.n_groups * worker / n_workers, grp_hi = P.n_groups * (worker + 1) / n_workers;
    if (grp_lo >= grp_hi) return;
    // a stretch that starts inside a clip: the group before it, without stores
    const bool warm = (grp_lo % P.groups_per_clip) != 0;
    for (int64_t grp = warm ? grp_lo - 1 : grp_lo; grp < grp_hi; ++grp) {
        const bool emit = grp >= grp_lo;
        const int64_t b = grp / P.groups_per_clip;
        const int t0 = (int)(grp - b * P.groups_per_clip) * 8;
        const bool live = t0 + g < Ti;
        const bool last = t0 + 8 >= Ti;                                         // last group of its clip
        const ApOutBuf sb = ap_outbuf_make(const_cast<char *>(reinterpret_cast<const char *>(P.S)) + b * clip_bytes, clip_bytes);
        ap_float2 v[R];
        apq_inverse_frame<R>(sb, live ? lane_bytes : 0xF0000000u, Tsi, t0, Sm + q * BS, Tt + q * BS, Ln, msrc, v);
        // Overlap-add into the accumulator in n_round = ceil(N / hop) rounds: frames n_round apart do not overlap,
        // so round r adds the frames g = r (mod n_round) with plain read-add-write (LDS float atomics cost ~150
        // cycles per wave instruction here: 1.0 ms of a 1.5 ms kernel).  8-byte accesses when hop is even.
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) v[k1] = ap_mul2(v[k1], wv[k1]);
        {
            float *dst = acc + g * hop + 2 * R * Ln.k2;
            const bool even_hop = (hop & 1) == 0;
            for (int r = 0; r < n_round; ++r) {
                if (live && g % n_round == r) {
                    if (even_hop) {
                        ap_float2 o[R];
#pragma unroll
                        for (int k1 = 0; k1 < R; ++k1) o[k1] = reinterpret_cast<ap_float2 *>(dst)[k1];
#pragma unroll
                        for (int k1 = 0; k1 < R; ++k1) reinterpret_cast<ap_float2 *>(dst)[k1] = ap_add(o[k1], v[k1]);
                    } else {
#pragma unroll
                        for (int k1 = 0; k1 < R; ++k1) { dst[2 * k1] += v[k1].x; dst[2 * k1 + 1] += v[k1].y; }
                    }
                }
                AP_WAVE_SYNC();
            }
        }
        AP_WAVE_SYNC();
        const int64_t p0 = (int64_t)t0 * hop;                                   // position of acc[0] in the un-trimmed signal
        if (emit) {
            float *yb = P.y + b * P.out_len;
            // position fq hop + sidx of the accumulator, fq = 0 .. 7 (the whole span for the clip's last group)
            const int n_fq = last ? (span + hop - 1) / hop : 8;
            for (int sidx = lane; sidx < hop; sidx += 64) {
                const float wss_all = WSS[sidx];
                for (int fq = 0; fq < n_fq; ++fq) {
                    const int j = fq * hop + sidx;
                    if (j >= span) break;
                    const int64_t i = p0 + j - P.out_offset;
                    if (i < 0 || i >= P.out_len) continue;
                    // window-sum-squares of the frames that exist and cover this position: frame t0 + fq - m reads
                    // the window at sidx + m hop
                    int lastf = t0 + fq;
                    float wss;
                    if (lastf <= Ti - 1 && (int64_t)p0 + j >= N - 1) {
                        wss = wss_all;
                    } else {
                        int m_lo = lastf > Ti - 1 ? lastf - (Ti - 1) : 0;          // frames past T do not exist
                        int m_hi = (N - 1 - sidx) / hop;                           // window index < N
                        if (m_hi > lastf) m_hi = lastf;                            // frames before 0 do not exist
                        wss = 0.0f;
                        for (int m = m_hi; m >= m_lo; --m) wss += W2[sidx + m * hop];   // increasing frame index
                    }
                    yb[i] = acc[j] / fmaxf(wss, 1e-8f);
                }
            }
            if (last) {                                                         // nothing covers the rest of the row
                for (int64_t i = p0 + span - P.out_offset + lane; i < P.out_len; i += 64)
                    if (i >= 0) yb[i] = 0.0f;
            }
        }
        AP_WAVE_SYNC();
        // the tail becomes the next group's carry (a new clip starts from zeros)
        float cr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = lane + 64 * u;
            cr[u] = (!last && j < n_carry) ? acc[8 * hop + j] : 0.0f;
        }
        AP_WAVE_SYNC();
        for (int j = lane; j < span; j += 64) acc[j] = 0.0f;
        AP_WAVE_SYNC();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = lane + 64 * u;
            if (j < n_carry) acc[j] = cr[u];
        }
        AP_WAVE_SYNC();
    }
}
