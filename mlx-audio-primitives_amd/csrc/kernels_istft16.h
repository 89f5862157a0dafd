// Fused ISTFT, n_fft = 2048, hop 256 / 512 / 1024: S (B, 1025, T) complex with rows `Ts` apart -> y (B, out_len)
// (transpose + mx.fft.irfft + overlap-add + trim, stft.py:292-338; overlap_add.metal:16-55) — round 3.
//
// The transform and the overlap-add are those of ap_irfft2048_wave_kernel<1> (kernels_wave.h: one frame per
// wave, 8 frames per step, the windowed frames gathered out of the waves' LDS buffers with a carry along the
// workgroup's contiguous stretch); what is new is how the spectrum comes in.  tools/load_probe.hip
// (profiles/r03_load_probe.txt) showed that for LOADS of this layout the alignment of a row segment does not
// matter and its length does (8 frames = 64 bytes: every 128-byte line is requested twice; FETCH_SIZE of the
// 8-frame kernel was 1.72 x the spectrum even with its sector-aligned windows and their 68 carry registers).  So a
// workgroup loads 16 FRAMES of every row at once, as they fall (16 lanes x 8 bytes = one or two lines, whole
// lines for the padded-row layout ap_stft_rows_f32 writes), one 16-frame group ahead of its use, stages them
// through LDS in 8 chunks of 128 bins so that every wave ends up with the bins of its TWO frames (t0 + w now,
// t0 + 8 + w kept in registers for the second step), and runs two 8-frame steps per load.
#pragma once
#include "kernels_wave.h"
#include "ap_phase_clock.h"


// HS = log2(hop): 8, 9 or 10 (the gather's loop bounds and shifts are compile-time constants)
// TIGHT: the fenced, register-lean form of the transform (kernels_wave.h).  The unfenced one also fits (233 VGPRs, no
// scratch) but measured 2.4 % slower here on the same box (0.3673 vs 0.3587 ms): AP_ISTFT16_LOOSE=1 selects it.
// SPREAD: the next group's loads are issued from inside the first step's transform, two chunks at a time between its
// stages, instead of in the staging pass.
// TILE: the staging pass moves the group in two rounds (chunks 0..5 over the exchange buffers + the staging area, then
// chunks 6, 7 and bin 512) with four barriers instead of eight rounds of one chunk.
template <int HS, bool TIGHT = true, int SPREAD = 0, int TILE = 0>
__global__ void __launch_bounds__(64 * APS_WAVES, 2) ap_istft2048_g16_kernel(ApIstft16Params P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APW_X_COMPLEX;
    const ap_float2 *TW2 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw2);
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);
    ap_float2 *IB = reinterpret_cast<ap_float2 *>(ap_smem + P.off_ib);        // [2][129][17]
    const int F = APW_NC + 1;
    const int Ti = (int)P.T, Ts = (int)P.Ts;
    // A workgroup owns a contiguous stretch of the (clip, 8-frame step) stream, [h_lo, h_hi): the loads come in
    // 16-frame groups (two steps), so a stretch may begin with the second step of a group (the first one is then
    // its store-less warm-up, from the same loads) or end with the first.  Steps rather than groups as the unit
    // keep the stretches of a small problem even (64 clips x 216 frames on 256 workgroups: 6.75 steps each).
    const int64_t hpc = (P.T + APS_WAVES - 1) / APS_WAVES;                       // steps per clip
    const int64_t n_h = hpc * (P.n_g16 / P.g16_per_clip);
    const int64_t h_lo = n_h * (int64_t)blockIdx.x / gridDim.x;
    const int64_t h_hi = n_h * ((int64_t)blockIdx.x + 1) / gridDim.x;
    if (h_lo >= h_hi) return;
    AP_PH_DECL();
    // A 16-frame group travels as one word, (clip << 32) | first frame, -1 = none; the loop below steps (clip, step in
    // clip) along instead of dividing (a 64-bit division by a run-time divisor is ~100 scalar instructions, and the
    // loaders used to make one per chunk)
    auto group_code = [](int64_t b, int t0) { return (b << 32) | (int64_t)(uint32_t)t0; };
    const int64_t b_lo = h_lo / hpc;
    const int hc_lo = (int)(h_lo - b_lo * hpc);                                  // first step's index in its clip
    // loader role: thread (sq = tid / 16, sf = tid % 16) fetches frame t0 + sf of the rows sq, 32 + sq (bins
    // 64 c + l) and 64 + sq, 96 + sq (bins 1024 - 64 c - l) of every chunk c, + bin 512 by the first 16 threads
    const int sq0 = tid >> 4, sf0 = tid & 15;
    ap_float2 pre[8][4], pre_mid = ap_mk(0.0f, 0.0f);
    // (32-bit arithmetic: the launch code bounds Ts by 2^20, so bin Ts + t < 2^31)
    // chunk c of the 16-frame group g16 -> pre[c] (c == 7: + bin 512)
    // The clip is a raw buffer resource: a load's address is base + lane offset (VGPR) + row offset (SGPR), so the 33
    // loads of a group cost two lane offsets instead of 33 64-bit address computations (those were ~300 of the VALU
    // instructions a wave spends per group, in a kernel whose transforms are VALU-bound), and a lane whose frame does
    // not exist parks its offset out of range and reads zeros.  Mirrored rows count up from row 993 - 64 c - 32 j
    // (lane part 31 - sq) so that no lane offset is negative.  The launch code bounds the clip by 0xF0000000 bytes.
    const int64_t clip_bytes = (int64_t)F * P.Ts * (int64_t)sizeof(ap_float2);
    auto load_chunk = [&](int64_t g16, int c) {
        const int64_t b = AP_UNIFORM((int)(g16 >> 32));           // (explicitly uniform: the clip's buffer resource is scalar)
        const int t = AP_UNIFORM((int)(uint32_t)g16) + sf0;
        const ApOutBuf sb = ap_outbuf_make(const_cast<char *>(reinterpret_cast<const char *>(P.S)) + b * clip_bytes, clip_bytes);
        const bool ok = t < Ti;
        const unsigned lp = ok ? 8u * (unsigned)(sq0 * Ts + t) : 0xF0000000u;
        const unsigned lm = ok ? 8u * (unsigned)((31 - sq0) * Ts + t) : 0xF0000000u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = i & 1;
            pre[c][i] = (i >> 1) ? ap_outbuf_load2(sb, lm, 8u * (unsigned)((APW_NC - 31 - 64 * c - 32 * j) * Ts))
                                 : ap_outbuf_load2(sb, lp, 8u * (unsigned)((64 * c + 32 * j) * Ts));
        }
        if (c == 7) pre_mid = ap_outbuf_load2(sb, ok && tid < APS16_G ? 8u * (unsigned)t : 0xF0000000u, 8u * (unsigned)((APW_NC / 2) * Ts));
    };
    auto load16 = [&](int64_t g16) {
#pragma unroll
        for (int c = 0; c < 8; ++c) load_chunk(g16, c);
    };
    // the stretch's first loads go out before the tables are built: they land under the set-up
    // an even first step inside a clip is preceded by the second step of the group before it (other loads); an odd
    // one by the first step of its own group
    const bool warm_prev_group = hc_lo > 0 && !(hc_lo & 1);
    const int64_t g_first = group_code(b_lo, (hc_lo >> 1) * APS16_G);
    load16(warm_prev_group ? group_code(b_lo, ((hc_lo >> 1) - 1) * APS16_G) : g_first);
    {
        ap_float2 *tw2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2);
        ap_float2 *tw1 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1);
        if (tid < 64)       // signs of the quad stage folded in, as in apw_fill_tables
            tw2[(tid >> 4) * 17 + (tid & 15)] = ap_scale(P.tw[32 * (tid >> 4) * (tid & 15)], ((tid >> 4) == 1 || (tid >> 4) == 2) ? -1.0f : 1.0f);
        for (int i = tid; i < 16 * 64; i += 64 * APS_WAVES) tw1[i] = P.tw[2 * (i & 63) * (i >> 6)];
        float *win = reinterpret_cast<float *>(ap_smem + P.off_win);
        for (int i = tid; i < 2 * APW_NC; i += 64 * APS_WAVES) win[i] = P.window[i];
    }
    const ApwLane lc = apw_lane_init(lane, TW2, P.tw);
    AP_LDS_BARRIER();
    {
        // 1 / max(sum of w^2 over the n_fft / hop frames that cover a position, 1e-8) by phase: inside a clip
        // (every covering frame exists) the divisor of overlap_add.metal:44-52 only depends on the position mod hop
        // (from the LDS copy of the window; visible to the gathers after the staging pass's barriers)
        const float *win = reinterpret_cast<const float *>(ap_smem + P.off_win);
        float *inv = reinterpret_cast<float *>(ap_smem + P.off_inv);
        for (int p = tid; p < P.hop; p += 64 * APS_WAVES) {
            float w2 = 0.0f;
            for (int q = p; q < 2 * APW_NC; q += P.hop) w2 += win[q] * win[q];
            inv[p] = 1.0f / fmaxf(w2, 1e-8f);
        }
    }
    const float scale = 1.0f / 1024.0f;    // 1/n_fft, and the merge below works at half scale
    // pre -> LDS -> every wave's bins of its two frames: xk[r] = S[64 r + lane], xm[r] = S[1024 - 64 r - lane]
    // `next16 >= 0`: as soon as chunk c has left its registers, the loads of chunk c of that group are issued
    // into them: they have a whole period (two transforms and two gathers) to land, and the bursts of the 256
    // workgroups (131 KB each) do not have to be served inside one gather.
    auto stage = [&](ap_float2 (&xkA)[8], ap_float2 (&xmA)[8], ap_float2 &xhA, ap_float2 (&xkB)[8], ap_float2 (&xmB)[8],
                     ap_float2 &xhB, int64_t next16) {
        if constexpr (TILE != 0) {
            // Two rounds.  The first one's 768 rows start at the exchange buffers (idle between the second step's gather
            // and the merge: the barrier below) and run on into the staging area, which follows them in LDS; the second
            // one's 257 rows are the staging area's alone, so the merge may start while other waves still read them.
            ap_float2 *T1 = reinterpret_cast<ap_float2 *>(ap_smem);
            ap_float2 *T2 = IB;
            AP_LDS_BARRIER();                                            // every wave has left the gather
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    T1[(128 * c + (i >> 1) * 64 + sq0 + 32 * (i & 1)) * APS16_OB_ROW + sf0] = pre[c][i];
            AP_PH(8);
            AP_LDS_BARRIER();
            AP_PH(9);
            if (!SPREAD && next16 >= 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) load_chunk(next16, c);
            }
            AP_PH(10);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                xkA[c] = T1[(128 * c + lane) * APS16_OB_ROW + wave];
                xkB[c] = T1[(128 * c + lane) * APS16_OB_ROW + 8 + wave];
                xmA[c] = T1[(128 * c + 64 + lane) * APS16_OB_ROW + wave];
                xmB[c] = T1[(128 * c + 64 + lane) * APS16_OB_ROW + 8 + wave];
            }
            AP_LDS_BARRIER();                                            // the staging area's first rows were round one's
#pragma unroll
            for (int c = 6; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    T2[(128 * (c - 6) + (i >> 1) * 64 + sq0 + 32 * (i & 1)) * APS16_OB_ROW + sf0] = pre[c][i];
            if (tid < APS16_G) T2[256 * APS16_OB_ROW + sf0] = pre_mid;
            AP_PH(8);
            AP_LDS_BARRIER();
            AP_PH(9);
            if (!SPREAD && next16 >= 0) {
                load_chunk(next16, 6);
                load_chunk(next16, 7);
            }
            AP_PH(10);
#pragma unroll
            for (int c = 6; c < 8; ++c) {
                xkA[c] = T2[(128 * (c - 6) + lane) * APS16_OB_ROW + wave];
                xkB[c] = T2[(128 * (c - 6) + lane) * APS16_OB_ROW + 8 + wave];
                xmA[c] = T2[(128 * (c - 6) + 64 + lane) * APS16_OB_ROW + wave];
                xmB[c] = T2[(128 * (c - 6) + 64 + lane) * APS16_OB_ROW + 8 + wave];
            }
            xhA = T2[256 * APS16_OB_ROW + wave];
            xhB = T2[256 * APS16_OB_ROW + 8 + wave];
            return;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ap_float2 *buf = IB + (c & 1) * (APS16_OB_ROWS * APS16_OB_ROW);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                buf[((i >> 1) * 64 + sq0 + 32 * (i & 1)) * APS16_OB_ROW + sf0] = pre[c][i];
            if (c == 7 && tid < APS16_G) buf[128 * APS16_OB_ROW + sf0] = pre_mid;
            AP_PH(8);
            AP_LDS_BARRIER();
            AP_PH(9);
            if (!SPREAD && next16 >= 0) load_chunk(next16, c);
            AP_PH(10);
            xkA[c] = buf[lane * APS16_OB_ROW + wave];
            xkB[c] = buf[lane * APS16_OB_ROW + 8 + wave];
            xmA[c] = buf[(64 + lane) * APS16_OB_ROW + wave];
            xmB[c] = buf[(64 + lane) * APS16_OB_ROW + 8 + wave];
            if (c == 7) {
                xhA = buf[128 * APS16_OB_ROW + wave];
                xhB = buf[128 * APS16_OB_ROW + 8 + wave];
            }
        }
    };

    constexpr int H = 1 << HS;                                   // = P.hop
    constexpr int CN = 2 * APW_NC - H;                           // carry length
    constexpr int hs = HS;
    const ap_float2 *WINP = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);   // (w[2n], w[2n+1])
    const float *WIN = reinterpret_cast<const float *>(ap_smem + P.off_win);
    const float *INV = reinterpret_cast<const float *>(ap_smem + P.off_inv);
    const float *XF = reinterpret_cast<const float *>(ap_smem);  // frame f at XF + f * 2 APW_X_COMPLEX
    constexpr int n_own = APS_WAVES * H;                         // positions one 8-frame step completes

    // One 8-frame step: frames t0 .. t0 + 7 of clip b from the waves' (xk, xm, xh); `half` picks the carry
    // buffers (steps alternate); `next16 >= 0`: issue the loads of that 16-frame group once xk / xm are consumed.
    auto step = [&](ap_float2 (&xk)[8], ap_float2 (&xm)[8], ap_float2 xh, int64_t b, int t0i, int half, bool emit,
                    int64_t next16) {
        // SPREAD: point pt = 0..3 of the group's first step (before the merge, before and after the transform, after the
        // windowed frame is out) issues chunks 2 pt, 2 pt + 1 of the next group
        auto issue = [&](int pt) __attribute__((always_inline)) {
            if (SPREAD && next16 >= 0) {
                AP_SCHED_FENCE();
                load_chunk(next16, 2 * pt);
                load_chunk(next16, 2 * pt + 1);
                AP_SCHED_FENCE();
            }
        };
        AP_PH(0);
        if (half) AP_LDS_BARRIER();
        AP_PH(6);
        AP_PRIO(3);
        issue(0);
        ap_float2 v[16];
        ap_float2 tws0h = lc.tws0h;          // opaque per step: keeps the 8 merge twiddles out of loop-invariant registers
        AP_PIN(tws0h.x);
        AP_PIN(tws0h.y);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            ap_float2 a_k = xk[r], a_m = xm[r];
            if (r == 0 && lane == 0) { a_k.y = 0.0f; a_m.y = 0.0f; }      // DC / Nyquist imaginary parts ignored
            const ap_float2 a = ap_add_conj(a_k, a_m);
            const ap_float2 d = ap_sub_conj(a_k, a_m);
            const ap_float2 w = r == 0 ? tws0h : ap_mul_bw_c(tws0h, APW_C32(r), APW_S32(r));
            const ap_float2 o = ap_mul_bw(d, w);                           // W^-k = (c, +s)
            v[r] = ap_fma_sub_swap(a, lc.halfc, o);                         // index lane + 64 r
            // index 1024 - k belongs to lane 64 - lane (register 15 - r): exchange through LDS
            const int km = (APW_NC - (lane + 64 * r)) & (APW_NC - 1);
            if (!(r == 0 && lane == 0)) X[apw_zidx(km)] = ap_fma_add_mi(a, lc.half, o);
        }
        if (lane == 0) X[apw_zidx(APW_NC / 2)] = xh;    // bin 512 pairs with itself: conj Z[512] / 2 = X[512]
        AP_WAVE_SYNC();
#pragma unroll
        for (int j = 8; j < 16; ++j) v[j] = X[apw_zidx(lane + 64 * j)];
        AP_WAVE_SYNC();
        AP_SCHED_FENCE();
        AP_PH(1);
        issue(1);
        apw_forward<false, TIGHT, true>(v, X, TW1, lc);
        AP_SCHED_FENCE();
        issue(2);
        AP_PH(2);
        // ---- fused overlap-add ---------------------------------------------------------------
        float *carry_in = reinterpret_cast<float *>(ap_smem + P.off_carry) + half * CN;
        float *carry_out = reinterpret_cast<float *>(ap_smem + P.off_carry) + (half ^ 1) * CN;
        {   // windowed frame -> this wave's exchange buffer (padded natural order: conflict-free)
            ap_float2 wv[16];
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) wv[cc] = WINP[lc.k1p + 16 * cc + 256 * lc.qd];
            AP_WAVE_SYNC();                                          // quad stage done with X
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) {
                const int n = lc.k1p + 16 * cc + 256 * lc.qd;
                X[apw_zidx(n)] = ap_mul2(ap_mul2(v[cc], ap_mk(scale, -scale)), wv[cc]);
            }
        }
        if (t0i == 0)                                                // a clip starts: nothing carried in
            for (int i = tid; i < CN; i += 64 * APS_WAVES) carry_in[i] = 0.0f;
        issue(3);
        AP_PRIO(0);
        AP_PH(3);
        AP_LDS_BARRIER();
        AP_PH(7);
        const bool clip_last = t0i + APS_WAVES >= Ti;
        const int64_t p0 = (int64_t)t0i * H;                         // padded position of r = 0
        float *yb = P.y + b * P.out_len;
        const int64_t n0 = p0 - P.out_offset;                        // output index of r = 0
        // A thread's positions r = 4 tid + 2048 k all lie in hop number rq = r >> hs = (wave >> (hs - 8)) + k (2048 >> hs),
        // the same for the whole wave: which frames cover r, whether it has a carry in or out, whether it is emitted and
        // which divisor applies are SCALAR decisions (spelled through rq so that the compiler makes them scalar branches
        // instead of compares, masks and divergent loops in every lane)
        constexpr int NQ = (2 * APW_NC) >> hs;                       // frames that cover a position (n_fft / hop)
        for (int r = 4 * tid;; r += 4 * 64 * APS_WAVES) {
            const int rq = AP_UNIFORM(r >> hs);
            if (rq >= APS_WAVES - 1 + NQ) break;                     // r >= n_own + CN
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            if (rq < NQ - 1) {                                       // r < CN
                const ap_float4 c4 = *reinterpret_cast<const ap_float4 *>(carry_in + r);
                s0 = c4.x; s1 = c4.y; s2 = c4.z; s3 = c4.w;
            }
            // frames of this step that cover r: f H <= r < f H + n_fft
            const int f_cov = rq - NQ + 1;                           // first covering frame, may be < 0
            const int f_lo = f_cov < 0 ? 0 : f_cov;
            const int f_hi = rq > APS_WAVES - 1 ? APS_WAVES - 1 : rq;
            for (int f = f_lo; f <= f_hi; ++f) {
                const int sidx = r - (f << hs);                      // sample index in frame f (multiple of 4)
                const ap_float4 q = *reinterpret_cast<const ap_float4 *>(
                    XF + f * (2 * APW_X_COMPLEX) + 2 * apw_zidx(sidx >> 1));
                s0 += q.x; s1 += q.y; s2 += q.z; s3 += q.w;
            }
            if (rq >= APS_WAVES) {                                   // r >= n_own: contributions to the next step's range
                ap_float4 c4; c4.x = s0; c4.y = s1; c4.z = s2; c4.w = s3;
                *reinterpret_cast<ap_float4 *>(carry_out + (r - n_own)) = c4;
            }
            if (emit && (rq < APS_WAVES || clip_last)) {
                // window-sum-of-squares over the frames of the CLIP that cover the position
                // (relative frame numbers; frames before this step count too)
                const int F_lo = f_cov < -t0i ? -t0i : f_cov;
                const int F_hi = rq > Ti - 1 - t0i ? Ti - 1 - t0i : rq;
                float w0, w1, w2, w3;                                // reciprocals of the divisors
                if (F_lo == f_cov && F_hi == rq) {                   // every covering frame exists: by phase
                    const ap_float4 iv = *reinterpret_cast<const ap_float4 *>(INV + (r & (H - 1)));
                    w0 = iv.x; w1 = iv.y; w2 = iv.z; w3 = iv.w;
                } else {                                             // the clip's first and last n_fft - hop positions
                    w0 = w1 = w2 = w3 = 0.0f;
                    for (int Fi = F_lo; Fi <= F_hi; ++Fi) {
                        const ap_float4 w = *reinterpret_cast<const ap_float4 *>(WIN + (r - Fi * H));
                        w0 += w.x * w.x; w1 += w.y * w.y; w2 += w.z * w.z; w3 += w.w * w.w;
                    }
                    w0 = 1.0f / fmaxf(w0, 1e-8f); w1 = 1.0f / fmaxf(w1, 1e-8f);
                    w2 = 1.0f / fmaxf(w2, 1e-8f); w3 = 1.0f / fmaxf(w3, 1e-8f);
                }
                const int64_t n = n0 + r;
                if (n >= 0 && n + 3 < P.out_len) {
                    ap_float4 o4;
                    o4.x = s0 * w0;
                    o4.y = s1 * w1;
                    o4.z = s2 * w2;
                    o4.w = s3 * w3;
                    if (((reinterpret_cast<uintptr_t>(yb + n)) & 15) == 0) {
                        *reinterpret_cast<ap_float4 *>(yb + n) = o4;
                    } else {
                        yb[n] = o4.x; yb[n + 1] = o4.y; yb[n + 2] = o4.z; yb[n + 3] = o4.w;
                    }
                } else {
                    if (n >= 0 && n < P.out_len) yb[n] = s0 * w0;
                    if (n + 1 >= 0 && n + 1 < P.out_len) yb[n + 1] = s1 * w1;
                    if (n + 2 >= 0 && n + 2 < P.out_len) yb[n + 2] = s2 * w2;
                    if (n + 3 >= 0 && n + 3 < P.out_len) yb[n + 3] = s3 * w3;
                }
            }
        }
        if (emit && clip_last) {                                     // no frame reaches beyond the tail: 0 / 1e-8
            int64_t n = p0 + n_own + CN - P.out_offset;
            if (n < 0) n = 0;
            for (n += tid; n < P.out_len; n += 64 * APS_WAVES) yb[n] = 0.0f;
        }
        AP_PH(4);
    };

    ap_float2 xkA[8], xmA[8], xhA, xkB[8], xmB[8], xhB;
    AP_PH(5);
    // A stretch that starts inside a clip first re-runs the 8 frames before it with the stores disabled: they
    // rebuild the carry (its length 2048 - hop is at most 7 frames for hop >= 256).
    if (warm_prev_group) {
        stage(xkA, xmA, xhA, xkB, xmB, xhB, g_first);
        step(xkB, xmB, xhB, b_lo, ((hc_lo >> 1) - 1) * APS16_G + 8, 1, false, g_first);
    }
    int64_t b = b_lo;
    int hc = hc_lo;
    for (int64_t h = h_lo; h < h_hi;) {
        const int t0 = (hc >> 1) * APS16_G;
        const bool startB = hc & 1;                                  // only the stretch's first step can be odd
        const bool haveB = t0 + 8 < Ti;                              // the clip has frames in the group's second half
        const bool ownB = haveB && (startB || h + 1 < h_hi);
        const int adv = startB ? 1 : (haveB ? 2 : 1);                // to the first step of the next group
        const int64_t h_next = h + adv;
        int hc_n = hc + adv;
        int64_t b_n = b;
        if (hc_n >= (int)hpc) { hc_n = 0; ++b_n; }
        const int64_t next16 = h_next < h_hi ? group_code(b_n, (hc_n >> 1) * APS16_G) : -1;
        stage(xkA, xmA, xhA, xkB, xmB, xhB, next16);
        step(xkA, xmA, xhA, b, t0, 0, !startB, next16);               // startB: the warm-up of the step this stretch starts with
        // (the carry parity continues either way: a clip that ends after a first step starts the next one with a
        //  zeroed carry_in of parity 0)
        if (ownB) step(xkB, xmB, xhB, b, t0 + 8, 1, true, -1);
        h = h_next;
        b = b_n;
        hc = hc_n;
    }
    AP_PH_FLUSH();
}
