// STFT, n_fft = 2048, complex output (B, 1025, T) with T fastest (stft.py:216) — 16 frames per group.
//
// Round 3.  tools/store_probe.hip (profiles/r03_store_probe.txt) measured what the row segments of
// this layout cost on MI355X, store-only, 905 MB: 64-byte segments as they fall 1.5 TB/s, 64-byte
// sector-aligned windows 3.1 TB/s (what ap_stft2048_wave_kernel emits), 128-byte LINE-aligned windows
// 4.1 TB/s (4.5 non-temporal), longer windows no better; a misaligned 128-byte segment 1.9 TB/s.  So a
// row has to leave as whole aligned 128-byte lines = 16 frames.
//
// The 8 waves of a workgroup transform 16 CONSECUTIVE frames, two each (wave w: frames t0 + w and
// t0 + 8 + w; the first one's spectrum waits in registers while the second is transformed with the
// same register / LDS transform as kernels_wave.h), then transpose through LDS in 8 double-buffered
// chunks of 128 bins x 16 frames (one LDS-only barrier per chunk) so that 16 adjacent lanes write the
// 128 contiguous bytes of one bin's row.
//
// ALIGNED = 0 (the reference's contiguous layout, row stride T): rows start at arbitrary 8-byte
// offsets, so row k is written in line-ALIGNED windows that lag the group by 16 - phi(k) frames,
// phi(k) = frames from t0 to the row's next 128-byte boundary: the thread that owns position j of a
// row's window keeps the not yet written frame of the previous group in a register ("carry", 33
// complex per thread), the workgroup walks a contiguous stretch of one clip's groups so the window
// is completed one group later by the same thread, and the carries are flushed at the end of a clip
// / of the stretch (the scheme of ap_stft2048_wave_kernel at twice the window).
// ALIGNED = 1 (row stride a multiple of 16 complex and a 128-byte aligned base: the workspace layout
// of the Griffin-Lim loop): every group's segment is a whole line, no carries.
#pragma once
#include "kernels_wave.h"
#include "ap_phase_clock.h"
#include "kernels_pointwise.h"

#ifdef AP_HOST_EMU
struct ap_f4v { float x, y, z, w; };
#define AP_STORE2(p, v, NT) (*(p) = (v))
#else
typedef float ap_f4v __attribute__((ext_vector_type(4)));        // two complex values as one 16-byte register quad
#define AP_STORE2(p, v, NT)                                      \
    do {                                                         \
        if (NT) __builtin_nontemporal_store((v), (p));           \
        else *(p) = (v);                                         \
    } while (0)
#endif

// GL = 1 (needs ALIGNED = 1): the Griffin-Lim projection rides in the store phase (griffinlim.py:156-178).  Every
// thread fetches the previous raw spectrum and the target magnitude of ITS 33 elements (99 registers) before
// the group's first transform - they land under the two transforms - and then stores, per element, the raw bin
// (next iteration's "previous") and   rebuilt = R' + m (R' - S unit(prev)),  R' = S unit(raw).
// The separate projection pass over five (B, F, T) arrays and the re-read of the raw spectrum are gone.
//
// T2 = 1 (needs ALIGNED = 1, GL = 0; AP_STFT16_T2=1, NOT the default): the whole group is transposed at once - the
// 16 x 1025 tile lives in LDS as a whole (rows 0..575 over the waves' exchange buffers, which are idle between two
// groups' transforms, rows 576..1024 beside them), three LDS-only barriers per group instead of eight, every thread
// ends up with its 33 elements in registers and stores them as one burst (buffer addressing: SGPR row offset + one lane
// offset).  Measured round 3 with tools/phase_clock.py (profiles/README.md): the LDS phase does shrink from ~11 700 to
// ~3 200 cycles per group, but a CU's one vector-memory pipe takes a store instruction per ~40 cycles and a load per
// ~12, a wave blocks while its instructions wait to be taken, and 34 stores + 32 sample loads per thread in a row block
// the waves for 12 000 - 20 000 cycles per group; in the eight-round form the same traffic drains between the rounds'
// LDS latencies.  0.318 ms against 0.275 ms.  (Two more forms - the two waves of a SIMD storing at different times, and
// the stores dribbled through the next group's transforms four at a time - were slower still, 0.39 and 0.31 ms.)
template <int PADGEN, int ALIGNED, int NT, int GL = 0, int T2 = 0>
__global__ void __launch_bounds__(64 * APS_WAVES, 2) ap_stft2048_g16_kernel(ApStft16Params P) {
    static_assert(!GL || ALIGNED, "the fused projection needs the line-padded layout");
    static_assert(!T2 || (ALIGNED && !GL), "the whole-group tile is the line-padded layout's");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APW_X_COMPLEX;
    const ap_float2 *TW2 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw2);
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);
    const ap_float2 *WIN = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);
    ap_float2 *OB = reinterpret_cast<ap_float2 *>(ap_smem + P.off_ob);        // [2][129][17]
    // E2 (with T2): both frames of a group are fetched ahead of the store burst of the group before (vector memory
    // operations are taken in order by one pipe per CU: a load issued after the stores queues behind them)
    constexpr bool E2 = T2 && !PADGEN;
    ap_float2 raw[16];
    ap_float2 raw2[16];                 // T2: the group's second frame, fetched with the first one (before any store of the group before)
    // frame t0 + wave + 8 * second of the group
    auto load_frame_to = [&](ap_float2 (&raw)[16], int64_t group, int second) __attribute__((always_inline)) {
        // (a group travels as (clip << 32) | first frame: the loop steps both along, no division per use)
        // (made uniform explicitly: the compiler does not see it through the packed word and builds the clip's buffer
        //  resource in a waterfall loop otherwise)
        const int64_t b = AP_UNIFORM((int)(group >> 32));
        const int64_t t = (int64_t)AP_UNIFORM((int)(uint32_t)group) + wave + 8 * second;
        const float *yb = P.y + b * P.L;
        const ApClip clip = ap_clip_make(yb, P.L);
        const int64_t base = t * (int64_t)P.hop - P.pad;          // wave-uniform
        // frames beyond T read past the clip: zeros (never stored).  Edge / reflect padding only
        // touches the frames that overlap a clip boundary.
        const bool inside = !PADGEN || t >= P.T || (base >= 0 && base + 2 * APW_NC <= P.L);
#ifdef AP_PHASE_CLOCK
        if (P.stagger & 0x200) {                  // diagnostic knock-out: no sample loads
#pragma unroll
            for (int j = 0; j < 16; ++j) raw[j] = ap_mk((float)lane, (float)j);
            return;
        }
#endif
        if (inside) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int64_t p = base + 2 * (lane + 64 * j);
                raw[j] = ap_clip_load2(clip, (int)p);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int64_t p = base + 2 * (lane + 64 * j);
                raw[j] = ap_mk(ap_load_padded(yb, P.L, p, P.pad_mode), ap_load_padded(yb, P.L, p + 1, P.pad_mode));
            }
        }
    };
    const int64_t g_lo = P.n_groups * (int64_t)blockIdx.x / gridDim.x;
    const int64_t g_hi = P.n_groups * ((int64_t)blockIdx.x + 1) / gridDim.x;
    auto load_frame = [&](int64_t group, int second) __attribute__((always_inline)) { load_frame_to(raw, group, second); };
    auto group_code = [](int64_t b, int64_t t0) { return (b << 32) | t0; };
    const int gpc = (int)P.groups_per_clip;
    int64_t b_cur = g_lo / P.groups_per_clip;                        // the stretch's first group: clip and index in the clip
    int gi_cur = (int)(g_lo - b_cur * P.groups_per_clip);
    if (!PADGEN && g_lo < g_hi) load_frame(group_code(b_cur, gi_cur * APS16_G), 0);        // the first frame's samples land under the table set-up
    if (E2 && g_lo < g_hi) load_frame_to(raw2, group_code(b_cur, gi_cur * APS16_G), 1);
    apw_fill_tables(reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2),
                    reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1),
                    reinterpret_cast<ap_float2 *>(ap_smem + P.off_win), P.tw, P.window, tid,
                    64 * APS_WAVES);
    const ApwLane lc = apw_lane_init(lane, TW2, P.tw);
    AP_LDS_BARRIER();

#ifndef AP_HOST_EMU
    // Every workgroup alternates a compute phase (two transforms) with a burst of 131 KB of stores, all 256 of
    // them with the same period: started together they also burst together, the memory system idles during the
    // transforms and the stores queue up behind each other during the bursts.  A start-up delay of a quarter
    // period per workgroup class spreads the bursts over the period.
    for (int d = (int)((blockIdx.x >> 3) & 3) * (P.stagger & 0xFF); d > 0; --d) __builtin_amdgcn_s_sleep(127);
#endif
    AP_PH_DECL();
    const int F = APW_NC + 1;
    const int sq0 = tid >> 4, sf0 = tid & 15;                                // store role of this thread
    const int Ts = (int)P.Ts, Ts15 = (int)(P.Ts & 15);
    ap_float2 carry[8][4], carry_mid = ap_mk(0.0f, 0.0f);
    if (!ALIGNED) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) carry[c][i] = ap_mk(0.0f, 0.0f);
    }
    // T2: this thread's 33 elements of a group (rows sq0 / 32 + sq0 of bins 64 c + l and 1024 - 64 c - l, + bin 512)
    ap_float2 r[8][4], r_mid = ap_mk(0.0f, 0.0f);
    auto issue_stores = [&](int64_t grp) {
        const int64_t gb = AP_UNIFORM((int)(grp >> 32));
        const int64_t gt0 = AP_UNIFORM((int)(uint32_t)grp);
        const int64_t first = gb * (int64_t)F * P.Ts + gt0;       // out[gb, 0, gt0]; the resource ends with the clip's last row
        const ApOutBuf od = ap_outbuf_make(P.out + first, ((int64_t)F * P.Ts - gt0) * (int64_t)sizeof(ap_float2));
        const bool ok = sf0 < (int)(P.T - gt0);                    // the frame exists (lanes past the clip's end park)
        const unsigned lp = ok ? 8u * (unsigned)(sq0 * Ts + sf0) : 0xF0000000u;
        const unsigned lm = ok ? 8u * (unsigned)((31 - sq0) * Ts + sf0) : 0xF0000000u;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = i & 1;
                if (i >> 1) ap_outbuf_store2(od, lm, 8u * (unsigned)((APW_NC - 31 - 64 * c - 32 * j) * Ts), r[c][i]);
                else ap_outbuf_store2(od, lp, 8u * (unsigned)((64 * c + 32 * j) * Ts), r[c][i]);
            }
        ap_outbuf_store2(od, ok && tid < APS16_G ? 8u * (unsigned)sf0 : 0xF0000000u, 8u * (unsigned)((APW_NC / 2) * Ts), r_mid);
    };
    // tile row rho = 128 c + 64 h + l (h = 0: bin 64 c + l, h = 1: bin 1024 - 64 c - l), rho = 1024: bin 512
    ap_float2 *TR1 = reinterpret_cast<ap_float2 *>(ap_smem);               // rows 0 .. 575 (over the exchange buffers)
    ap_float2 *TR2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_ob);    // rows 576 .. 1024
    auto tile_row = [&](int c, int h, int l) -> ap_float2 * {             // c, h compile-time
        const int rho = 128 * c + 64 * h;
        return rho < APS16_T2_SPLIT ? TR1 + (rho + l) * APS16_OB_ROW : TR2 + (rho - APS16_T2_SPLIT + l) * APS16_OB_ROW;
    };

    // windowed frame in raw[] -> xk[r] = X[lane + 64 r], xm[r] = X[1024 - lane - 64 r], zh = Z[512]
    auto transform = [&](ap_float2 (&xk)[8], ap_float2 (&xm)[8], ap_float2 &zh, int64_t next_group, int next_second,
                         ap_float2 (&raw)[16]) __attribute__((always_inline)) {
        ap_float2 v[16];
        AP_SCHED_FENCE();
        // window in four batches: the 16 table values never sit in registers together (this is where the
        // first frame's spectrum, the prefetched samples and the carries meet)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ap_float2 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = WIN[lane + 64 * (4 * q + j)];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * q + j] = ap_mul2(raw[4 * q + j], w[j]);
            AP_SCHED_FENCE();
        }
        AP_PH(0);
        // (requested here, behind the last store rounds' backlog, the loads block the wave for 3 500 - 6 000 cycles; requested
        //  after the transform instead they do not land in time: 0.287-0.301 ms against 0.270-0.279, same box)
        if (!PADGEN && next_group >= 0) load_frame(next_group, next_second);   // lands under the transform
        AP_SCHED_FENCE();
        AP_PH(1);
        apw_forward<true, (!ALIGNED || GL)>(v, X, TW1, lc);
        AP_SCHED_FENCE();
        AP_PH(2);
        ApwLane ls = lc;
        if (!ALIGNED || GL) {                // keeps the 8 split twiddles from being hoisted out of the group loop
            AP_PIN(ls.tws0h.x);
            AP_PIN(ls.tws0h.y);
        }
        apw_split<true>(X, ls, xk, xm, zh);
        AP_SCHED_FENCE();
        AP_PH(3);
    };

    // every workgroup owns a contiguous stretch of the (clip, 16-frame group) stream: a row's
    // window is completed one group later by the same thread (ALIGNED = 0), and the lines of
    // consecutive groups follow each other from the same CU

    for (int64_t group = g_lo; group < g_hi; ++group, b_cur = gi_cur + 1 >= gpc ? b_cur + 1 : b_cur, gi_cur = gi_cur + 1 >= gpc ? 0 : gi_cur + 1) {
        // (GL = 1 sits at 256 registers: the stepped form cost it 8 spilled ones, so it keeps the division)
        const int64_t b = GL ? group / P.groups_per_clip : b_cur;
        const int64_t t0 = GL ? (group - b * P.groups_per_clip) * APS16_G : (int64_t)gi_cur * APS16_G;
        const int64_t gc = group_code(b, t0);                                       // this group and the next one
        const int64_t gc_n = t0 + APS16_G >= (int64_t)gpc * APS16_G ? group_code(b + 1, 0) : group_code(b, t0 + APS16_G);
        ap_float2 *ob = P.out + b * (int64_t)F * P.Ts + t0;
        // (complex index of out[b, 0, t0]) mod 16: the same for every group of a clip
        const int a0 = (int)(((reinterpret_cast<uintptr_t>(P.out) >> 3) + (uint64_t)(b * (int64_t)F * P.Ts + t0)) & 15);
        const bool have_prev = group > g_lo && t0 > 0;          // carries hold this clip's previous group
        const bool last = group + 1 == g_hi || t0 + APS16_G >= P.T;
        const int trem = (int)(P.T - t0);                        // frames t0 + i with i < trem exist

        ap_float2 xkA[8], xmA[8], zhA, xkB[8], xmB[8], zhB;
        ap_float2 gpv[8][4], gpv_mid = ap_mk(0.0f, 0.0f);        // GL: previous raw spectrum of this thread's elements
        float gmg[8][4], gmg_mid = 0.0f;                          // GL: target magnitudes
        // GL = 2: the store role is (row sq8 = tid / 8, frames 2 sp and 2 sp + 1, sp = tid % 8) of the chunk's rows sq8
        // (bin 64 c + sq8) and 64 + sq8 (bin 1024 - 64 c - sq8): the same 33 elements per thread as 16-BYTE accesses to
        // the three complex arrays.  The CU's vector-memory pipe charges per instruction (~18 cycles a store, ~12-16 a
        // load whatever its width: tools/store_width_probe.hip, ta_probe.hip): 34 stores + 17 wide loads per thread and
        // group instead of 66 + 33.  Measured: bit-identical and no faster (5.31-5.35 ms against 5.23-5.27 for 32
        // iterations at cfg3) - the waves wait for the memory system behind the pipe, not for the pipe; AP_GL_PAIRS=1.
        const int sq8 = tid >> 3, sp2 = 2 * (tid & 7);
        ap_f4v gpw[8][2], gpw_mid = {0.0f, 0.0f, 0.0f, 0.0f};
        float gmw[8][2][2], gmw_mid0 = 0.0f, gmw_mid1 = 0.0f;
        const bool ok0 = sp2 < trem, ok1 = sp2 + 1 < trem;       // the thread's two frames exist
        auto gl2_bin = [&](int c, int e, int q) __attribute__((always_inline)) { return e ? APW_NC - 64 * c - q : 64 * c + q; };
        auto gl2_load_mag = [&](int c0, int c1) __attribute__((always_inline)) {
            int q = sq8, f = sp2;
            AP_PIN(q);
            AP_PIN(f);
            const float *mb = P.gl_mag + b * (int64_t)F * P.T + t0;
            const int Td = (int)P.T;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c >= c0 && c < c1) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        gmw[c][e][0] = ok0 ? mb[gl2_bin(c, e, q) * Td + f] : 0.0f;
                        gmw[c][e][1] = ok1 ? mb[gl2_bin(c, e, q) * Td + f + 1] : 0.0f;
                    }
                    if (c == 7 && tid < 8) {
                        gmw_mid0 = ok0 ? mb[(APW_NC / 2) * Td + f] : 0.0f;
                        gmw_mid1 = ok1 ? mb[(APW_NC / 2) * Td + f + 1] : 0.0f;
                    }
                }
        };
        // the magnitudes of chunks c0 .. c1 - 1 (the first two ride with the previous spectrum before the
        // transforms, the rest are fetched when the transforms' registers are free again: 25 registers less at the peak)
        auto gl_load_mag = [&](int c0, int c1) {
            int sq = sq0, sf = sf0;
            AP_PIN(sq);
            AP_PIN(sf);
            const float *mb = P.gl_mag + b * (int64_t)F * P.T + t0;
            const bool ok = sf < trem;
            const int Td = (int)P.T;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c >= c0 && c < c1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int l = sq + 32 * (i & 1);
                        const int bin = (i >> 1) ? APW_NC - 64 * c - l : 64 * c + l;
                        gmg[c][i] = ok ? mb[bin * Td + sf] : 0.0f;
                    }
                    if (c == 7 && tid < APS16_G) gmg_mid = ok ? mb[(APW_NC / 2) * Td + sf] : 0.0f;
                }
        };
        if (GL == 2) {
            // (rows are padded to whole lines: a pair whose second frame does not exist still reads inside its row)
            int q = sq8, f = sp2;
            AP_PIN(q);
            AP_PIN(f);
            const ap_float2 *pb = P.gl_prev + b * (int64_t)F * P.Ts + t0;
            const ap_f4v zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    gpw[c][e] = ok0 ? *reinterpret_cast<const ap_f4v *>(&pb[gl2_bin(c, e, q) * Ts + f]) : zero4;
            if (tid < 8) gpw_mid = ok0 ? *reinterpret_cast<const ap_f4v *>(&pb[(APW_NC / 2) * Ts + f]) : zero4;
        } else if (GL) {
            int sq = sq0, sf = sf0;
            AP_PIN(sq);
            AP_PIN(sf);
            const ap_float2 *pb = P.gl_prev + b * (int64_t)F * P.Ts + t0;
            const bool ok = sf < trem;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int l = sq + 32 * (i & 1);
                    const int bin = (i >> 1) ? APW_NC - 64 * c - l : 64 * c + l;
                    gpv[c][i] = ok ? pb[bin * Ts + sf] : ap_mk(0.0f, 0.0f);
                }
            if (tid < APS16_G) gpv_mid = ok ? pb[(APW_NC / 2) * Ts + sf] : ap_mk(0.0f, 0.0f);
        }
        if (PADGEN) load_frame(gc, 0);
        transform(xkA, xmA, zhA, E2 ? -1 : gc, 1, raw);
        if (GL) {                                  // the first four chunks' magnitudes land under the second transform
            AP_SCHED_FENCE();
            if (GL == 2) gl2_load_mag(0, 4);
            else gl_load_mag(0, 4);
            AP_SCHED_FENCE();
        }
        if (PADGEN) load_frame(gc, 1);
        transform(xkB, xmB, zhB, -1, 0, E2 ? raw2 : raw);
        // the next group's first frame lands under the store phase (during the second transform the
        // registers hold the first frame's spectrum instead)
        AP_SCHED_FENCE();
        if (GL == 2) gl2_load_mag(4, 8);
        else if (GL) gl_load_mag(4, 8);
        // (GL: the projection's operands fill the registers until half of the chunks are out: the prefetch waits)
        if (!GL && !PADGEN && !T2 && group + 1 < g_hi) load_frame(gc_n, 0);
        AP_SCHED_FENCE();
        AP_PH(4);

        if constexpr (T2 != 0) {
            AP_PH(8);
            // the part of the tile beside the exchange buffers first; the rest once every wave has left its transform
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const bool second = 128 * c + 64 * h >= APS16_T2_SPLIT;
                        if (second != (pass == 0)) continue;
                        ap_float2 *row = tile_row(c, h, lane);
                        row[wave] = h ? xmA[c] : xkA[c];
                        row[8 + wave] = h ? xmB[c] : xkB[c];
                    }
                if (pass == 0 && lane == 0) {                            // X[512] = conj Z[512]
                    ap_float2 *row = TR2 + (1024 - APS16_T2_SPLIT) * APS16_OB_ROW;
                    row[wave] = ap_mk(zhA.x, -zhA.y);
                    row[8 + wave] = ap_mk(zhB.x, -zhB.y);
                }
                AP_LDS_BARRIER();
            }
            AP_PH(5);
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) r[c][i] = tile_row(c, i >> 1, sq0 + 32 * (i & 1))[sf0];
            r_mid = TR2[(1024 - APS16_T2_SPLIT) * APS16_OB_ROW + sf0];
            AP_LDS_BARRIER();                                            // the exchange buffers are the waves' again
            AP_PH(6);
            if (!PADGEN && group + 1 < g_hi) {                      // into an idle pipe, ahead of the stores
                load_frame(gc_n, 0);
                load_frame_to(raw2, gc_n, 1);
            }
            issue_stores(gc);
            AP_PH(7);
            continue;
        }

        // ---- transposed store: chunk c holds bins 64 c + lane and 1024 - 64 c - lane ----------
        // (the thread's store role is made opaque per group: otherwise the compiler hoists the row
        //  offsets of all 33 elements out of the group loop and spills the carries to make room)
        int sq = sq0, sf = sf0;
        AP_PIN(sq);
        AP_PIN(sf);
        // ALIGNED = 0: phi(bin) = frames from t0 to row `bin`'s next 128-byte boundary = -(a0 + bin Ts) mod 16.  The rows
        // a thread owns are the bins 64 c + l and 1024 - 64 c - l with l = sq or sq + 32, and 64 Ts, 32 Ts and 1024 Ts
        // are multiples of 16: phi takes THREE values per thread and group (k = 0: -(a0 + sq Ts), k = 1: -(a0 - sq Ts),
        // k = 2: bin 512, -a0), and with it the staging column to read, whether the position is stored from the carry,
        // where, and whether it is stored at all - once per group instead of once per element (the per-element form
        // was 310 of this kernel's 810 VALU instructions per frame).
        int k_rd[3];
        bool k_take[3], k_store[3], k_flush[3];
        ap_float2 *k_ob[3];
        if (!ALIGNED) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ph = (0 - (a0 + (k == 0 ? sq * Ts15 : k == 1 ? -sq * Ts15 : 0))) & 15;
                k_rd[k] = (sf + ph) & 15;
                k_take[k] = sf < 16 - ph;
                const int dt = ph + sf - 16;                       // frame t0 + dt of the row
                k_store[k] = k_take[k] ? have_prev : dt < trem;
                k_flush[k] = k_take[k] && ph + sf < trem;          // at a flush the carry just taken is frame t0 + dt + 16
                k_ob[k] = ob + dt;
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ap_float2 *buf = OB + (c & 1) * (APS16_OB_ROWS * APS16_OB_ROW);
            buf[lane * APS16_OB_ROW + wave] = xkA[c];
            buf[lane * APS16_OB_ROW + 8 + wave] = xkB[c];
            buf[(64 + lane) * APS16_OB_ROW + wave] = xmA[c];
            buf[(64 + lane) * APS16_OB_ROW + 8 + wave] = xmB[c];
            if (c == 7 && lane == 0) {                             // X[512] = conj Z[512]
                buf[128 * APS16_OB_ROW + wave] = ap_mk(zhA.x, -zhA.y);
                buf[128 * APS16_OB_ROW + 8 + wave] = ap_mk(zhB.x, -zhB.y);
            }
            AP_PH(5);
            AP_LDS_BARRIER();
            AP_PH(6);
            if (GL && c == 4) {
                AP_SCHED_FENCE();
                if (!PADGEN && group + 1 < g_hi) load_frame(gc_n, 0);
                AP_SCHED_FENCE();
            }
            // thread (sq = tid / 16, sf = tid % 16) owns position sf of the windows of the chunk's rows
            // sq, 32 + sq (bins 64 c + l) and 64 + sq, 96 + sq (bins 1024 - 64 c - l), l = sq, 32 + sq
            int bins[5], slots[5];
            const int ne = c == 7 ? 5 : 4;                         // + bin 512 in the last chunk
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int l = sq + 32 * (i & 1);
                bins[i] = (i >> 1) ? APW_NC - 64 * c - l : 64 * c + l;
                slots[i] = (i >> 1) * 64 + l;
            }
            bins[4] = APW_NC / 2;
            slots[4] = 128;
            if constexpr (GL == 2) {
                // two frames of two rows (+ bin 512 in the last chunk, threads 0..7) as 16-byte stores
                int q = sq8, f = sp2;
                AP_PIN(q);
                AP_PIN(f);
                const int ne2 = c == 7 ? 3 : 2;
                ap_float2 x0[3], x1[3];
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (e < ne2) {
                        const int slot = e == 2 ? 128 : e * 64 + q;
                        x0[e] = buf[slot * APS16_OB_ROW + f];
                        x1[e] = buf[slot * APS16_OB_ROW + f + 1];
                    }
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (e < ne2) {
                        const bool mine = e < 2 || tid < 8;
                        const int bin = e == 2 ? APW_NC / 2 : gl2_bin(c, e, q);
                        const float s0m = e == 2 ? gmw_mid0 : gmw[c][e == 2 ? 0 : e][0], s1m = e == 2 ? gmw_mid1 : gmw[c][e == 2 ? 0 : e][1];
                        const ap_f4v pv = e == 2 ? gpw_mid : gpw[c][e == 2 ? 0 : e];
                        // the arithmetic of ap_gl_rows_kernel (kernels_pointwise.h), operation for operation
                        const ap_float2 u0 = ap_unit_phase(x0[e]), u1 = ap_unit_phase(x1[e]);
                        ap_float2 r0 = ap_mk(s0m * u0.x, s0m * u0.y), r1 = ap_mk(s1m * u1.x, s1m * u1.y);
                        if (P.gl_momentum > 0.0f) {
                            const ap_float2 v0 = ap_unit_phase(ap_mk(pv.x, pv.y)), v1 = ap_unit_phase(ap_mk(pv.z, pv.w));
                            r0 = ap_mk(r0.x + P.gl_momentum * (r0.x - s0m * v0.x), r0.y + P.gl_momentum * (r0.y - s0m * v0.y));
                            r1 = ap_mk(r1.x + P.gl_momentum * (r1.x - s1m * v1.x), r1.y + P.gl_momentum * (r1.y - s1m * v1.y));
                        }
                        ap_float2 *po = &ob[bin * Ts + f];
                        ap_float2 *pr = &P.gl_rebuilt[b * (int64_t)F * P.Ts + t0 + bin * Ts + f];
                        if (mine && ok1) {
                            const ap_f4v w4 = {x0[e].x, x0[e].y, x1[e].x, x1[e].y};
                            const ap_f4v r4 = {r0.x, r0.y, r1.x, r1.y};
                            *reinterpret_cast<ap_f4v *>(po) = w4;
                            *reinterpret_cast<ap_f4v *>(pr) = r4;
                        } else if (mine && ok0) {
                            *po = x0[e];
                            *pr = r0;
                        }
                    }
            } else if (ALIGNED) {
                ap_float2 x[5];
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < ne) x[i] = buf[slots[i] * APS16_OB_ROW + sf];
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < ne) {
                        const bool mine = i < 4 || tid < APS16_G;
#ifdef AP_PHASE_CLOCK
                        if (P.stagger & 0x100) { if (x[i].x == 123.456f) ob[0] = x[i]; continue; }   // diagnostic knock-out: no stores
#endif
                        if (mine && sf < trem) AP_STORE2(&ob[bins[i] * Ts + sf], x[i], NT);
                        if (GL == 1 && mine && sf < trem) {
                            // the arithmetic of ap_gl_rows_kernel (kernels_pointwise.h), operation for operation
                            const float sm = i < 4 ? gmg[c][i] : gmg_mid;
                            const ap_float2 u = ap_unit_phase(x[i]);
                            ap_float2 rn = ap_mk(sm * u.x, sm * u.y);
                            if (P.gl_momentum > 0.0f) {
                                const ap_float2 v = ap_unit_phase(i < 4 ? gpv[c][i] : gpv_mid);
                                rn = ap_mk(rn.x + P.gl_momentum * (rn.x - sm * v.x), rn.y + P.gl_momentum * (rn.y - sm * v.y));
                            }
                            P.gl_rebuilt[b * (int64_t)F * P.Ts + t0 + bins[i] * Ts + sf] = rn;
                        }
                    }
            } else {
                // One LDS read per element: frame (sf + phi) mod 16 of this group is either stored now
                // (sf >= 16 - phi) or becomes the carry while the old carry is stored.  phi only depends on the
                // element's class (cls[]: bins 64 c + l, bins 1024 - 64 c - l, bin 512 - see the group's head).
                ap_float2 x[5];
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < ne) {                                  // all LDS reads of the chunk first
                        const int k = i == 4 ? 2 : (i >> 1);
                        x[i] = buf[slots[i] * APS16_OB_ROW + k_rd[k]];
                    }
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < ne) {
                        const int k = i == 4 ? 2 : (i >> 1);
                        ap_float2 &cy = i < 4 ? carry[c][i] : carry_mid;
                        const bool mine = i < 4 || tid < APS16_G;
                        const ap_float2 val = k_take[k] ? cy : x[i];
                        if (mine && k_store[k]) AP_STORE2(&k_ob[k][bins[i] * Ts], val, NT);
                        if (k_take[k]) cy = x[i];
                    }
                if (last) {                                        // flush: the carries just taken
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        if (i < ne) {
                            const int k = i == 4 ? 2 : (i >> 1);
                            const bool mine = i < 4 || tid < APS16_G;
                            if (mine && k_flush[k]) AP_STORE2(&k_ob[k][bins[i] * Ts + 16], x[i], NT);
                        }
                }
            }
            AP_PH(7);
        }
    }
    AP_PH_FLUSH();
}

// (Round 3 also measured a software-pipelined form of the ALIGNED kernel: the transform cut at its two wave-private LDS
//  hand-offs into three resumable stages, and the eight store rounds of group g placed between the stages of group
//  g + 1's two transforms, so that a round's latencies sit under a stage's arithmetic.  0.335-0.344 ms against 0.285 ms
//  for the phase-by-phase kernel above on the same box: every round is a workgroup barrier, and eight barriers spread
//  through the transforms make the waves wait for each other's stages.  Not in the build; profiles/README.md.)
