// Fused mel-spectrogram for n_fft = 2048, the run kernel: 8 independent waves per CU, each wave
// one frame at a time in registers + its own LDS exchange buffer, output runs in registers.
//
// Same transform and contraction as ap_mel2048_wave_kernel (kernels_wave.h: 16 x 16 x 4 complex
// transform of the packed frame, paired real split, |X|^p plane, plan-based banded contraction),
// re-cut around what the round-2 measurements showed (profiles/README.md, tools/diag_clock.py,
// tools/power_probe.py):
//
//   * the chip runs this kernel at its POWER limit (1 340-1 355 W of board power, shader clock
//     2.2-2.35 GHz and falling whenever the pipes are kept busier): what shortens a launch is
//     less energy per frame - fewer instructions, fewer LDS and L2 bytes - not more waves (a
//     12-wave, 168-VGPR build spilled; alternating priorities between the co-resident waves
//     evened out their speeds and lowered the clock by the same factor);
//   * so everything frame-invariant is a template parameter or a register: the number of
//     contraction passes (NPASS), the hop class (HOPJ), (clip, frame) advanced incrementally
//     instead of a 64-bit division per frame, part descriptors as ready-made LDS addresses, rows
//     >= n_mels computed like the others and simply not stored, the window pairs and the split
//     twiddles of the lane in registers for the whole kernel (46 VGPRs: 16 LDS reads and 14
//     packed multiplies fewer per frame);
//   * the quad radix-4 runs on v_fmac_f32_dpp: the DPP operand feeds the FMA directly
//     (h += c * quad_perm(h)) and the two per-lane signs are folded into the W_64 twiddles that
//     precede it - 64 DPP FMAs + 48 plain ops instead of 64 DPP moves + 64 packed ops;
//   * the 8-frame output run of a lane's two mel rows lives in registers (32 contiguous bytes per
//     row and lane, as in ap_mel1024_wave_kernel): no output tile in LDS; the partial sums
//     alias the idle upper half of the wave's exchange buffer.
//
// Serves constant padding / center=False, power 2 or 1, n_mels <= 128, plans of <= 256 entries
// whose rows have <= 4 parts; everything else stays on ap_mel2048_wave_kernel.
// Reference: mel.py:245-352 (stft.py:130 + mel.py:344-350).
#pragma once
#include <type_traits>
#include "kernels_wave.h"

#define APM_WAVES 8           // waves per workgroup (2 per SIMD, 256 VGPRs each)
#define APM_RUN 8             // frames per output run held in registers
#define APM_PARTIAL_OFF 1152  // float offset of the partial sums inside the wave's X buffer (plane: 1025 floats)

// (ApmLane, apm_quad8 / apm_quad_radix4: kernels_wave.h - the STFT and irfft kernels use them too)

// Which frame-invariant per-lane tables stay in REGISTERS for the whole kernel instead of being
// re-read from LDS every frame (bit mask; 8 waves per CU leave 256 VGPRs per lane).  The product
// uses APM_REGS: with the two twiddle tables as well hipcc's scheduler spills inside the frame loop.
#define APM_REG_WIN 1         // window pairs                 32 VGPRs, saves 16 ds_read_b64 per frame
#define APM_REG_TW1 2         // W_1024^(lane k1)             30 VGPRs, saves 15
#define APM_REG_TW2 4         // (s1 s2) W_64^(a c)           30 VGPRs, saves 15
#define APM_REG_SPLIT 8       // W_2048^(lane + 64 r) / 2     14 VGPRs, saves 14 packed multiplies
#define APM_REGS (APM_REG_WIN | APM_REG_SPLIT)

// forward transform of apw_forward with the fmac-DPP quad stage; tw2row holds sg * W_64^(a c)
// (the samples x and the window pairs w go in separately: the window rides on the first butterfly level)
template <int REGS>
AP_DEV void apm_forward(const ap_float2 (&x)[16], const ap_float2 (&w)[16], ap_float2 *X, const ap_float2 *TW1,
                        const ap_float2 *tw2row, const ap_float2 (&t1r)[16], const ap_float2 (&t2r)[16],
                        const ApwLane &c, const ApmLane &m) {
    const int lane = c.lane;
    ap_float2 v[16];
    if (REGS & APM_REG_TW1) {
        ApButterfly<16>::run_weighted(x, w, v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = ap_mul_fw(v[k], t1r[k]);
    } else {
        ap_float2 t1[16];
#pragma unroll
        for (int k = 1; k < 16; ++k) t1[k] = TW1[k * 64 + lane];
        ApButterfly<16>::run_weighted(x, w, v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = ap_mul_fw(v[k], t1[k]);
    }
    {
        const int a = lane & 3, bq = lane >> 2;
#pragma unroll
        for (int k = 0; k < 16; ++k) X[APW_T1(k * 4 + a) + bq] = v[k];
    }
    AP_WAVE_SYNC();
    ap_float2 t2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = X[APW_T1(lane) + i];
    if (!(REGS & APM_REG_TW2)) {
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) t2[cc] = tw2row[cc];
    }
    AP_WAVE_SYNC();
    ApButterfly<16>::run(v);
    v[0] = ap_scale(v[0], m.sg);
#pragma unroll
    for (int cc = 1; cc < 16; ++cc) v[cc] = ap_mul_fw(v[cc], (REGS & APM_REG_TW2) ? t2r[cc] : t2[cc]);
    apm_quad_radix4(v, m);
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) X[apw_zidx(c.k1p + 16 * cc + 256 * c.qd)] = v[cc];
    AP_WAVE_SYNC();
}

// apw_split (kernels_wave.h) for |X| only, with the eight split twiddles W_2048^(lane + 64 r) / 2
// optionally held in registers
template <int REGS>
AP_DEV void apm_split(const ap_float2 *X, const ApwLane &c, const ap_float2 (&wsp)[8], ap_float2 (&xk)[8],
                      ap_float2 (&xm)[8], ap_float2 &zh) {
    ap_float2 zk[8], zm[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = c.lane + 64 * r;
        zk[r] = X[apw_zidx(k)];
        zm[r] = X[apw_zidx((APW_NC - k) & (APW_NC - 1))];
    }
    zh = X[apw_zidx(APW_NC / 2)];
    AP_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const ap_float2 a = ap_add_conj(zk[r], zm[r]);
        const ap_float2 d = ap_sub_conj(zk[r], zm[r]);
        const ap_float2 w = (REGS & APM_REG_SPLIT) ? wsp[r]
                                                   : (r == 0 ? c.tws0h : ap_mul_bw_c(c.tws0h, APW_C32(r), APW_S32(r)));
        const ap_float2 u = ap_mul_fw(d, w);
        xk[r] = ap_fma_add_mi(a, c.half, u);
        xm[r] = ap_fma_sub_mi(a, c.half, u);
    }
}

#ifdef AP_DIAG_STAMPS
// Diagnostic build only (tools/diag_clock.py; buffer in kernels_wave.h)
#define AP_DIAG_BEGIN() unsigned long long ap_dt0, ap_dr0; \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ap_dt0), "=s"(ap_dr0)::"memory")
#define AP_DIAG_END(widx) do { unsigned long long ap_dt1, ap_dr1; \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ap_dt1), "=s"(ap_dr1)::"memory"); \
    if ((threadIdx.x & 63) == 0 && (widx) < 256 * 16) { ap_diag_stamps[4 * (widx)] = ap_dt1 - ap_dt0; ap_diag_stamps[4 * (widx) + 1] = ap_dr1 - ap_dr0; ap_diag_stamps[4 * (widx) + 2] = ap_dr0; ap_diag_stamps[4 * (widx) + 3] = ap_dr1; } } while (0)
#else
#define AP_DIAG_BEGIN() do {} while (0)
#define AP_DIAG_END(widx) do {} while (0)
#endif

// The two (three) waves of a SIMD do not share it evenly: issue is arbitrated by priority, then AGE,
// so the first-dispatched wave of every SIMD ran its 54 frames in 144 us and the second in 186 us
// (tools/diag_clock.py), and the kernel lasts as long as the slow one.  The later-dispatched waves
// therefore raise their priority on every other frame: over two frames each side wins once, all
// waves finish together and the SIMDs stay shared until the end.  (At the power limit this buys
// nothing by itself - the clock drops as the SIMDs stay busier - but it keeps the tail short
// whenever the kernel is not power-bound: short batches, cooler boards.)
#ifdef AP_HOST_EMU
#define AP_FAIR_SHARE(wave, nw, f) do {} while (0)
#else
#define AP_FAIR_SHARE(wave, nw, f)                                   \
    do {                                                             \
        if ((wave) >= (nw) / 2 + ((nw) > 8 ? 2 : 0)) {               \
            if ((f) & 1) __builtin_amdgcn_s_setprio(1);              \
            else __builtin_amdgcn_s_setprio(0);                      \
        } else if ((nw) > 8 && (wave) >= 4) {                        \
            if ((f) & 1) __builtin_amdgcn_s_setprio(0);              \
            else __builtin_amdgcn_s_setprio(1);                      \
        }                                                            \
    } while (0)
#endif

template <int PMODE, int NPASS, int HOPJ, int IN16 = 0, int NW = APM_WAVES, int REGS = APM_REGS>
__global__ void __launch_bounds__(64 * NW, NW / 4) ap_mel2048_run_kernel(ApMelWaveParams P) {
    static_assert(IN16 != 1 || (REGS & APM_REG_WIN), "the PCM scale rides on the register-resident window");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APW_X_COMPLEX;
    ap_float2 *TW2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2);               // [4][17], signed
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);   // [16][64]
    const ap_float2 *WIN = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);   // [1024] pairs
    const ap_float4 *WQ = reinterpret_cast<const ap_float4 *>(ap_smem + P.off_wq);     // [n_quads]
    float *pp = reinterpret_cast<float *>(X);                 // |X|^p plane of this wave (floats 0..1024)
    float *partial = pp + APM_PARTIAL_OFF;                    // partial sums: idle part of X during the contraction
    const int M = P.n_mels;

    // ---------------- workgroup tables in LDS (once; the only workgroup barrier) ------
    {
        const int nt = 64 * NW;
        ap_float4 *wq = reinterpret_cast<ap_float4 *>(ap_smem + P.off_wq);
        for (int i = tid; i < P.n_quads; i += nt) wq[i] = reinterpret_cast<const ap_float4 *>(P.quads)[i];
        apw_fill_tables(TW2, reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1),
                        reinterpret_cast<ap_float2 *>(ap_smem + P.off_win), P.tw, P.window, tid, nt);
    }
    AP_LDS_BARRIER();
    // (apw_fill_tables has folded the quad stage's signs s1 s2 into the W_64 rows)
    const ApwLane lc = apw_lane_init(lane, TW2, P.tw);
    const ApmLane lm = apm_lane_init(lane);
    // per-lane tables kept in registers (REGS): straight from the global tables
    ap_float2 winr[16], t1r[16], t2r[16], wsp[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        winr[j] = (REGS & APM_REG_WIN) ? reinterpret_cast<const ap_float2 *>(P.window)[lane + 64 * j] : ap_mk(0.0f, 0.0f);
        if (IN16 == 1) winr[j] = ap_scale(winr[j], 1.0f / 32768.0f);      // 16-bit PCM -> [-1, 1): folded into the window
        t1r[j] = (REGS & APM_REG_TW1) ? P.tw[(2 * lane * j) & 2047] : ap_mk(0.0f, 0.0f);        // W_1024^(lane j)
        t2r[j] = (REGS & APM_REG_TW2) ? ap_scale(P.tw[32 * (lane & 3) * j], lm.sg) : ap_mk(0.0f, 0.0f);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
        wsp[r] = (REGS & APM_REG_SPLIT) ? ap_scale(P.tw[lane + 64 * r], 0.5f) : ap_mk(0.0f, 0.0f);
    // frame-invariant contraction state: LDS addresses of this lane's entries
    const ap_float4 *pqa[NPASS], *pqb[NPASS];
    float *sa[NPASS], *sb[NPASS];
    float fz[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int pi = lane + 64 * ps;
        ap_int4 pd;
        pd.x = P.n_slots; pd.y = 0; pd.z = P.n_slots; pd.w = 0;          // idle entry: zero weights, dump slot
        if (pi < P.n_parts) pd = reinterpret_cast<const ap_int4 *>(P.parts)[pi];
        pqa[ps] = reinterpret_cast<const ap_float4 *>(pp) + pd.y;
        pqb[ps] = reinterpret_cast<const ap_float4 *>(pp) + pd.w;
        sa[ps] = partial + pd.x;
        sb[ps] = partial + (pd.z < 0 ? P.n_slots : pd.z);
        fz[ps] = pd.z < 0 ? 1.0f : 0.0f;                                  // one long part: both halves to slot A
    }
    int rs0[2], cnt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = lane + 64 * i;
        rs0[i] = row < M ? P.rowstart[row] : 0;
        cnt[i] = row < M ? P.rowstart[row + 1] - rs0[i] : 0;
    }
    AP_LDS_BARRIER();

    const int64_t worker = (int64_t)blockIdx.x * NW + wave;
    const int64_t n_workers = (int64_t)gridDim.x * NW;
    const int64_t n_frames = P.n_clips * P.T;
    const int64_t f_lo = n_frames * worker / n_workers, f_hi = n_frames * (worker + 1) / n_workers;
    const int Ti = (int)P.T;
    float vmax = -INFINITY;
    AP_DIAG_BEGIN();
    if (f_lo < f_hi) {
        int64_t b = f_lo / P.T;                   // the only division: (clip, frame) advance incrementally
        int t = (int)(f_lo - b * P.T);
        // IN16: P.y points at int16 samples (the ingest conversion rides on the
        // sample loads; half the HBM read bytes of the float32 path)
        const int16_t *y16 = reinterpret_cast<const int16_t *>(P.y);
        // IN16 == 2: float32 samples with reflect / edge padding (or odd hops): the frames that reach over a clip
        // end take the index-remapping loader, every other frame the bounds-checked loads
        ApClip clip = ap_clip_make(P.y + (IN16 == 1 ? 0 : b * P.L), IN16 == 1 ? 0 : P.L);
        ApClip16 clip16 = ap_clip16_make(y16 + (IN16 == 1 ? b * P.L : 0), IN16 == 1 ? P.L : 0);
        auto ld2 = [&](int64_t bb, int base, int p) -> ap_float2 {
            if (IN16 == 1) return ap_clip16_load2(clip16, p);
            if (IN16 == 2 && !(base >= 0 && (int64_t)base + 2 * APW_NC <= P.L)) {
                const float *yb = P.y + bb * P.L;
                return ap_mk(ap_load_padded(yb, P.L, p, P.pad_mode), ap_load_padded(yb, P.L, p + 1, P.pad_mode));
            }
            return ap_clip_load2(clip, p);
        };
        // Sample pair j of the frame in hand lives in raw[(j + HOPJ rot) & 15]: with hop = 128 HOPJ, pair j of
        // frame t + 1 is pair j + HOPJ of frame t, so the HOPJ new pairs of the next frame overwrite the HOPJ
        // oldest registers and nothing moves.  `rot` has to be a compile-time constant for that (registers
        // cannot be indexed), hence a loop trip of U = 16 / HOPJ frames, one copy of the body per rotation.
        // (The earlier form shifted the 12 surviving pairs every frame: 24 v_mov_b64, 5 % of the issue slots.)
        constexpr int U = HOPJ == 4 ? 4 : 1;
        ap_float2 raw[16];
        auto load_frame = [&](int tt, auto rot_tag) {
            constexpr int ROT = decltype(rot_tag)::value;
            const int base = tt * P.hop - P.pad;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int p = base + 2 * (lane + 64 * j);
                raw[(j + HOPJ * ROT) & 15] = ld2(b, base, p);
            }
        };
        // U = 4: the stretch's first frame takes the register rotation r0 = t mod 4 and the run half (t / 4) mod 2, so
        // that frame t sits at position t mod 8 of the run registers: full runs are then the clip's frames 8 k .. 8 k + 7
        // and leave as ALIGNED 32-byte pieces of their rows (rows `Ts` floats apart; with Ts a multiple of 8 the
        // pieces are whole sectors - the output's write amplification was 1.9 x with runs that started wherever the
        // stretch did).  A clip that ends inside the stretch breaks the alignment for the rest of it.
        const int r0 = U == 4 ? (t & 3) : 0;
        if (r0 == 1) load_frame(t, std::integral_constant<int, 1 % U>());
        else if (r0 == 2) load_frame(t, std::integral_constant<int, 2 % U>());
        else if (r0 == 3) load_frame(t, std::integral_constant<int, 3 % U>());
        else load_frame(t, std::integral_constant<int, 0>());
        // The run of finished frames (rows lane and lane + 64) waits in registers for a 32-byte store:
        // position p8 = 4 h + rot of an 8-frame cycle (U = 4: static register, uniform half h), or a
        // shift register (U = 1).
        float acc0[APM_RUN], acc1[APM_RUN];
#pragma unroll
        for (int i = 0; i < APM_RUN; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
        int nrun = 0, half = U == 4 ? ((t >> 2) & 1) : 0;
        int64_t f = f_lo;

        // one frame; returns false after the last frame of the stretch
        auto frame = [&](auto rot_tag) -> bool {
            constexpr int ROT = decltype(rot_tag)::value;
            constexpr int NROT = (ROT + 1) % U;
            AP_FAIR_SHARE(wave, NW, f);
            const bool clip_ends = t + 1 == Ti;
            const bool more = f + 1 < f_hi;
            {
                ap_float2 xs[16], ws[16];         // renamings: no instructions
#pragma unroll
                for (int j = 0; j < 16; ++j) xs[j] = raw[(j + HOPJ * ROT) & 15];
                if (REGS & APM_REG_WIN) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) ws[j] = winr[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) ws[j] = WIN[lane + 64 * j];
                }
                AP_SCHED_FENCE();
                apm_forward<REGS>(xs, ws, X, TW1, lc.tw2row, t1r, t2r, lc, lm);
            }
            // The next frame of this wave's stretch, in flight during split + contraction
            AP_SCHED_FENCE();
            if (more) {
                if (clip_ends) {                  // next clip starts
                    if (IN16 == 1) clip16 = ap_clip16_make(y16 + (b + 1) * P.L, P.L);
                    else clip = ap_clip_make(P.y + (b + 1) * P.L, P.L);
                }
                // hop = 128 HOPJ: only the HOPJ new pairs are loaded, the shared samples stay where they are;
                // a new clip (and HOPJ = 0) loads the other pairs as well, into the same registers.  (One
                // code path on purpose: with the full load in a branch of its own the register allocator
                // copied all 12 surviving pairs out and back on every frame.)
                const int base = (clip_ends ? 0 : t + 1) * P.hop - P.pad;
#pragma unroll
                for (int j = 16 - HOPJ; j < 16; ++j) {
                    const int p = base + 2 * (lane + 64 * j);
                    raw[(j + HOPJ * NROT) & 15] = ld2(clip_ends ? b + 1 : b, base, p);
                }
                if (HOPJ == 0 || clip_ends) {
#pragma unroll
                    for (int j = 0; j < 16 - HOPJ; ++j) {
                        const int p = base + 2 * (lane + 64 * j);
                        raw[(j + HOPJ * NROT) & 15] = ld2(clip_ends ? b + 1 : b, base, p);
                    }
                }
            }
            AP_SCHED_FENCE();
            {
                ap_float2 xk[8], xm[8], zh;
                apm_split<REGS>(X, lc, wsp, xk, xm, zh);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int k = lane + 64 * r;
                    pp[k] = apw_pow2x<PMODE>(xk[r].x, xk[r].y, P.power);
                    pp[APW_NC - k] = apw_pow2x<PMODE>(xm[r].x, xm[r].y, P.power);
                }
                if (lane == 0) pp[APW_NC / 2] = apw_pow2x<PMODE>(zh.x, zh.y, P.power);
            }
            AP_WAVE_SYNC();
            // ---- plan-based contraction: NPASS branch-free passes (kernels_wave.h) -------------
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const ap_float4 *wq = WQ + 256 * ps + lane;
                ap_float4 w[4], q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = wq[64 * i];
                q[0] = pqa[ps][0]; q[1] = pqa[ps][1]; q[2] = pqb[ps][0]; q[3] = pqb[ps][1];
                float acc[2] = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i >> 1] = fmaf(w[i].x, q[i].x, acc[i >> 1]);
                    acc[i >> 1] = fmaf(w[i].y, q[i].y, acc[i >> 1]);
                    acc[i >> 1] = fmaf(w[i].z, q[i].z, acc[i >> 1]);
                    acc[i >> 1] = fmaf(w[i].w, q[i].w, acc[i >> 1]);
                }
                *sa[ps] = fmaf(fz[ps], acc[1], acc[0]);
                *sb[ps] = acc[1];
            }
            AP_WAVE_SYNC();
            // ---- row sums (<= 4 adjacent slots per row) into the run's registers ---------------
            float sum2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float p0 = partial[rs0[i]], p1 = partial[rs0[i] + 1], p2 = partial[rs0[i] + 2],
                            p3 = partial[rs0[i] + 3];
                float sum = cnt[i] > 0 ? p0 : 0.0f;
                sum += cnt[i] > 1 ? p1 : 0.0f;
                sum += cnt[i] > 2 ? p2 : 0.0f;
                sum += cnt[i] > 3 ? p3 : 0.0f;
                sum2[i] = sum;
                vmax = fmaxf(vmax, cnt[i] > 0 ? sum : vmax);
            }
            // the frame's position in the run registers, and the position of the run's first frame
            int pos;
            if (U == 1) {
#pragma unroll
                for (int i = 0; i < APM_RUN - 1; ++i) { acc0[i] = acc0[i + 1]; acc1[i] = acc1[i + 1]; }
                acc0[APM_RUN - 1] = sum2[0];
                acc1[APM_RUN - 1] = sum2[1];
                pos = APM_RUN - 1;
            } else {
                if (half) { acc0[4 + ROT] = sum2[0]; acc1[4 + ROT] = sum2[1]; }        // uniform branch
                else { acc0[ROT] = sum2[0]; acc1[ROT] = sum2[1]; }
                pos = 4 * half + ROT;
            }
            ++nrun;
            AP_WAVE_SYNC();
            // ---- store the run when it is full, the clip ends or the stretch ends ---------------
            if ((U == 1 ? nrun == APM_RUN : pos == APM_RUN - 1) || clip_ends || !more) {
                // frame t - nrun + 1 + g sits in register pos - nrun + 1 + g
                const int first = pos - nrun + 1;
                float *ob = P.out + b * (int64_t)M * P.Ts + (t - nrun + 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = lane + 64 * i;
                    if (row < M) {
                        float *dst = ob + (int64_t)row * P.Ts;
                        const float *src = i == 0 ? acc0 : acc1;
                        if (nrun == APM_RUN) {      // 32 contiguous bytes: two 16-byte stores (4-byte aligned)
                            ap_rsp_f4u lo, hi;
                            lo.x = src[0]; lo.y = src[1]; lo.z = src[2]; lo.w = src[3];
                            hi.x = src[4]; hi.y = src[5]; hi.z = src[6]; hi.w = src[7];
                            *reinterpret_cast<ap_rsp_f4u *>(dst) = lo;
                            *reinterpret_cast<ap_rsp_f4u *>(dst + 4) = hi;
                        } else {
#pragma unroll
                            for (int g = 0; g < APM_RUN; ++g)
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
    AP_DIAG_END((int)worker);
    if (P.max_key) {                  // one atomic per wave: lanes -> LDS -> lane 0
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

// ---------------------------------------------------------------------------------------------
// Spectral centroid / bandwidth / rolloff / flatness of every frame straight from the audio, n_fft = 2048
// (reference features.py:57-442: an STFT, |.|, |.|^p and sum / cumsum / argmax chains per call).  The
// transform, the paired split and the |X|^p plane are those of the mel run kernel above; instead of the
// filterbank contraction the wave reduces its frame's 1025 plane values: lane l owns the 16 contiguous
// bins 16 l .. 16 l + 15 (lane 63 bin 1024 as well), sums them, and the 64 lane sums are combined with
// wave shuffles; the rolloff lane is found from an exclusive scan of the lane sums.  Nothing but the
// samples is read from HBM and 4-16 bytes per frame are written: the complex spectrum (8.2 KB per frame,
// written and read back by the two-kernel route of kernels_features.h) never exists.
// ---------------------------------------------------------------------------------------------
#ifdef AP_HOST_EMU
AP_DEV float apm_lane_xor(float x, int mask) { return emu_lane_xor(x, mask); }
AP_DEV float apm_lane_up(float x, int d, int lane) { return emu_lane_perm(x, lane - d); }
#else
AP_DEV float apm_lane_xor(float x, int mask) { return __shfl_xor(x, mask, 64); }
AP_DEV float apm_lane_up(float x, int d, int lane) { return __shfl_up(x, d, 64); }
#endif
AP_DEV float apm_wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += apm_lane_xor(x, off);
    return x;
}

// FLAT: flatness wanted (17 logs per lane and frame); PGEN: bandwidth exponent p != 2 (powf in the loop -
// a template flag so that the usual p = 2 build stays inside the instruction cache)
template <int PMODE, int HOPJ, int FLAT, int PGEN = 0, int NW = APM_WAVES, int REGS = APM_REGS>
__global__ void __launch_bounds__(64 * NW, NW / 4) ap_spec2048_run_kernel(ApSpecWaveParams P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APW_X_COMPLEX;
    ap_float2 *TW2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2);
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);
    const ap_float2 *WIN = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);
    float *pp = reinterpret_cast<float *>(X);                 // |X|^p plane of this wave (floats 0..1024)
    apw_fill_tables(TW2, reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1),
                    reinterpret_cast<ap_float2 *>(ap_smem + P.off_win), P.tw, P.window, tid, 64 * NW);
    AP_LDS_BARRIER();
    // (apw_fill_tables has folded the quad stage's signs s1 s2 into the W_64 rows)
    const ApwLane lc = apw_lane_init(lane, TW2, P.tw);
    const ApmLane lm = apm_lane_init(lane);
    ap_float2 winr[16], t1r[16], t2r[16], wsp[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        winr[j] = (REGS & APM_REG_WIN) ? reinterpret_cast<const ap_float2 *>(P.window)[lane + 64 * j] : ap_mk(0.0f, 0.0f);
        t1r[j] = (REGS & APM_REG_TW1) ? P.tw[(2 * lane * j) & 2047] : ap_mk(0.0f, 0.0f);
        t2r[j] = (REGS & APM_REG_TW2) ? ap_scale(P.tw[32 * (lane & 3) * j], lm.sg) : ap_mk(0.0f, 0.0f);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
        wsp[r] = (REGS & APM_REG_SPLIT) ? ap_scale(P.tw[lane + 64 * r], 0.5f) : ap_mk(0.0f, 0.0f);
    float fk[17];                                             // bin centres of this lane's bins
#pragma unroll
    for (int i = 0; i < 16; ++i) fk[i] = P.freq[16 * lane + i];
    fk[16] = lane == 63 ? P.freq[APW_NC] : 0.0f;
    AP_LDS_BARRIER();

    const int64_t worker = (int64_t)blockIdx.x * NW + wave;
    const int64_t n_workers = (int64_t)gridDim.x * NW;
    const int64_t n_frames = P.n_clips * P.T;
    const int64_t f_lo = n_frames * worker / n_workers, f_hi = n_frames * (worker + 1) / n_workers;
    const int Ti = (int)P.T;
    if (f_lo >= f_hi) return;
    int64_t b = f_lo / P.T;
    int t = (int)(f_lo - b * P.T);
    ApClip clip = ap_clip_make(P.y + b * P.L, P.L);
    constexpr int U = HOPJ == 4 ? 4 : 1;                      // see ap_mel2048_run_kernel
    ap_float2 raw[16];
    auto load_frame = [&](int tt, auto rot_tag) {
        constexpr int ROT = decltype(rot_tag)::value;
        const int base = tt * P.hop - P.pad;
#pragma unroll
        for (int j = 0; j < 16; ++j) raw[(j + HOPJ * ROT) & 15] = ap_clip_load2(clip, base + 2 * (lane + 64 * j));
    };
    load_frame(t, std::integral_constant<int, 0>());
    int64_t f = f_lo;
    const float F_all = (float)(APW_NC + 1);

    auto frame = [&](auto rot_tag) -> bool {
        constexpr int ROT = decltype(rot_tag)::value;
        constexpr int NROT = (ROT + 1) % U;
        const bool clip_ends = t + 1 == Ti;
        const bool more = f + 1 < f_hi;
        {
            ap_float2 xs[16], ws[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) xs[j] = raw[(j + HOPJ * ROT) & 15];
            if (REGS & APM_REG_WIN) {
#pragma unroll
                for (int j = 0; j < 16; ++j) ws[j] = winr[j];
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) ws[j] = WIN[lane + 64 * j];
            }
            AP_SCHED_FENCE();
            apm_forward<REGS>(xs, ws, X, TW1, lc.tw2row, t1r, t2r, lc, lm);
        }
        AP_SCHED_FENCE();
        if (more) {
            if (clip_ends) clip = ap_clip_make(P.y + (b + 1) * P.L, P.L);
            const int base = (clip_ends ? 0 : t + 1) * P.hop - P.pad;
#pragma unroll
            for (int j = 16 - HOPJ; j < 16; ++j)
                raw[(j + HOPJ * NROT) & 15] = ap_clip_load2(clip, base + 2 * (lane + 64 * j));
            if (HOPJ == 0 || clip_ends) {
#pragma unroll
                for (int j = 0; j < 16 - HOPJ; ++j)
                    raw[(j + HOPJ * NROT) & 15] = ap_clip_load2(clip, base + 2 * (lane + 64 * j));
            }
        }
        AP_SCHED_FENCE();
        {
            ap_float2 xk[8], xm[8], zh;
            apm_split<REGS>(X, lc, wsp, xk, xm, zh);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = lane + 64 * r;
                pp[k] = apw_pow2x<PMODE>(xk[r].x, xk[r].y, P.power);
                pp[APW_NC - k] = apw_pow2x<PMODE>(xm[r].x, xm[r].y, P.power);
            }
            if (lane == 0) pp[APW_NC / 2] = apw_pow2x<PMODE>(zh.x, zh.y, P.power);
        }
        AP_WAVE_SYNC();
        // ---- this lane's 16 (17) contiguous bins -------------------------------------------
        float v[17];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const ap_float4 a = reinterpret_cast<const ap_float4 *>(pp)[4 * lane + q4];
            v[4 * q4] = a.x; v[4 * q4 + 1] = a.y; v[4 * q4 + 2] = a.z; v[4 * q4 + 3] = a.w;
        }
        v[16] = lane == 63 ? pp[APW_NC] : 0.0f;
        AP_WAVE_SYNC();                           // the plane is free for the next frame's transform
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 17; ++i) { s0 += v[i]; s1 = fmaf(fk[i], v[i], s1); }
        const float tot = apm_wave_sum(s0), tf = apm_wave_sum(s1);
        const float cen = tf / (tot + 1e-10f);
        const int64_t o = b * P.T + t;
        if (P.centroid && lane == 0) P.centroid[o] = cen;
        if (FLAT) {                               // features.py:427-437: exp(mean log max(S, amin)) / mean max(S, amin)
            float sl = 0.0f, sa = 0.0f;
#pragma unroll
            for (int i = 0; i < 17; ++i) {
                const float c = fmaxf(v[i], P.amin);
                if (i < 16 || lane == 63) { sl += logf(c); sa += c; }
            }
            const float tl = apm_wave_sum(sl), ta = apm_wave_sum(sa);
            if (P.flatness && lane == 0) P.flatness[o] = expf(tl / F_all) / (ta / F_all + 1e-10f);
        }
        if (P.bandwidth) {                        // features.py:242-266
            float dev = 0.0f;
#pragma unroll
            for (int i = 0; i < 17; ++i) {
                const float d = fabsf(fk[i] - cen);
                dev = fmaf(v[i], PGEN ? powf(d, P.p) : d * d, dev);
            }
            float w = apm_wave_sum(dev);
            if (P.norm) w = w / (tot + 1e-10f);
            if (lane == 0) P.bandwidth[o] = PGEN ? powf(w, 1.0f / P.p) : sqrtf(w);
        }
        if (P.rolloff) {                          // features.py:342-360: first bin whose running sum reaches the threshold
            float incl = s0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float up = apm_lane_up(incl, d, lane);
                if (lane >= d) incl += up;
            }
            const float before = incl - s0;
            const float thr = tot * P.roll_percent;
            const bool mine = (before + s0 >= thr) && (lane == 0 || before < thr);
            float run = before, fsel = lane == 63 ? fk[16] : fk[15];
            bool found = false;
#pragma unroll
            for (int i = 0; i < 17; ++i) {
                if (i < 16 || lane == 63) {
                    run += v[i];
                    if (!found && run >= thr) { found = true; fsel = fk[i]; }
                }
            }
            // lowest candidate of the wave (the tree-ordered scan is monotone only up to rounding, so there
            // may be none or two); no candidate: the last bin; sums that never reach thr (NaNs): bin 0
            float cand = (mine && found) ? fsel : INFINITY;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cand = fminf(cand, apm_lane_xor(cand, off));
            const float last = apm_lane_xor(fk[16], 63);               // lane 0 gets lane 63's bin 1024
            if (lane == 0) P.rolloff[o] = !(tot >= thr) ? fk[0] : (cand < INFINITY ? cand : last);
        }
        if (clip_ends) { t = 0; ++b; } else { ++t; }
        ++f;
        return more;
    };
    if (U == 1) {
        while (frame(std::integral_constant<int, 0>())) {}
    } else {
        for (;;) {
            if (!frame(std::integral_constant<int, 0>())) break;
            if (!frame(std::integral_constant<int, 1 % U>())) break;
            if (!frame(std::integral_constant<int, 2 % U>())) break;
            if (!frame(std::integral_constant<int, 3 % U>())) break;
        }
    }
}
