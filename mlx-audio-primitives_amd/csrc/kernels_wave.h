// Wave-per-frame kernels for n_fft = 2048 on gfx950: fused mel-spectrogram, STFT, irfft / ISTFT.
//
// Mel: every wave64 is an independent worker that owns a contiguous stretch of the flattened
// (clip, frame) stream and transforms ONE frame at a time entirely in its own registers + a
// wave-private LDS exchange buffer (no s_barrier anywhere in the frame loop):
//
//   global samples (bounds-checked buffer loads, one frame ahead; the samples two consecutive
//   frames share stay in registers) * window (LDS table)
//   radix-16 butterflies in registers, packed f32                  [mx.fft.rfft, stft.py:130]
//   * W_1024^(lane*k1) (LDS table) -> LDS transpose #1 (conflict-free rows, ds_read_b128)
//   radix-16 butterflies, * W_64^(a*c) (small LDS table)
//   radix-4 across the 4 lanes of a quad with DPP quad_perm (no LDS round trip)
//   LDS transpose #2 to natural order -> paired real-input split (bins k and 1024-k
//   share their sums) -> |X|^p -> one float plane per wave in LDS
//
// then the same wave contracts its plane with the mel filterbank (mx.matmul(mel_basis, S),
// mel.py:344-350) from a host-built plan: every filter's span is cut into parts of <= 4
// aligned 4-bin groups; a lane entry holds one long part or two short ones, reads 4 weight quads
// + 4 |X|^p quads without a branch, and the partial sums of a row (adjacent slots) are added by
// the lane that owns the row and stored as one column of the wave's output tile.  No atomics
// (LDS float atomics cost ~3 cycles per lane on gfx950 and made the first version of this
// kernel LDS-bound) and no workgroup barrier after the table set-up: the 8 waves of a workgroup
// only share the read-only tables (window, twiddles, filter weights) in LDS.
//
// The complex 1024-point transform is 16 x 16 x 4.  Workgroups are persistent (one per CU).
#pragma once
#include "ap_wave_params.h"
#include "fft_lds.h"
#include "kernels_generic.h"

#if defined(AP_DIAG_STAMPS) && !defined(AP_HOST_EMU)
// Diagnostic build only (tools/diag_clock.py): every wave of the n_fft=2048 mel kernels records how
// long its frame loop took in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime) ->
// the clock the chip held under this load.  Never compiled into the product library.
__device__ unsigned long long ap_diag_stamps[4 * 256 * 16];
#endif

// cos/sin(2 pi r / 32), r = 0..7, as compile-time constants (W_2048^(64 r) = W_32^r)
#define APW_C32(r) ((float)__builtin_cos(6.283185307179586476925 * (r) / 32.0))
#define APW_S32(r) ((float)__builtin_sin(6.283185307179586476925 * (r) / 32.0))

#ifdef AP_HOST_EMU
#define AP_WAVE_SYNC() emu_wave_sync()
#define AP_SCHED_FENCE() do {} while (0)
#define AP_PIN(x) do {} while (0)
#else
#define AP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// the value has to exist here: keeps a pure computation from being sunk past a later branch (where it
// would stretch the live ranges of its inputs across that branch)
#define AP_PIN(x) asm volatile("" : "+v"(x))
// wave-private LDS hand-off: the DS unit executes one wave's instructions in order, so
// only the compiler has to be kept from moving LDS accesses across this point.
#define AP_WAVE_SYNC()                                          \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

AP_DEV float ap_quad_xor1(float x) {   // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
AP_DEV float ap_quad_xor2(float x) {   // value of lane ^ 2 (quad_perm [2,3,0,1])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
}
#endif


// natural-order index k -> padded LDS slot: conflict-free ds_write_b64 of the quad
// outputs and (nearly) conflict-free ds_read_b64 of Z[lane+64r] / Z[1024-lane-64r]
// (tools/lds_banks.py)
AP_DEV int apw_zidx(int k) { return k + 4 * (k >> 8); }

// |X|^p with the exponent class fixed at compile time (2: power 2, 1: power 1,
// 0: anything else) so no powf code sits in the power-2 instruction stream.
template <int PMODE>
AP_DEV float apw_pow2x(float re, float im, float power) {
    const float p2 = re * re + im * im;
    if (PMODE == 2) return p2;
    if (PMODE == 1) return sqrtf(p2);
    return powf(sqrtf(p2), power);
}

#ifdef AP_HOST_EMU
#define AP_UNIFORM(x) (x)
// one clip as a bounds-checked buffer: out-of-range samples read as 0 (= constant padding)
struct ApClip { const float *base; int64_t n; };
AP_DEV ApClip ap_clip_make(const float *base, int64_t n) { ApClip c; c.base = base; c.n = n; return c; }
AP_DEV float ap_clip_load(const ApClip &c, int64_t idx) { return (idx >= 0 && idx < c.n) ? c.base[idx] : 0.0f; }
AP_DEV ap_float2 ap_clip_load2(const ApClip &c, int64_t idx) { return ap_mk(ap_clip_load(c, idx), ap_clip_load(c, idx + 1)); }
// 16-bit PCM clip: samples idx, idx + 1 as floats (unscaled integers), 0 outside [0, n)
struct ApClip16 { const int16_t *base; int64_t n; };
AP_DEV ApClip16 ap_clip16_make(const int16_t *base, int64_t n) { ApClip16 c; c.base = base; c.n = n; return c; }
AP_DEV ap_float2 ap_clip16_load2(const ApClip16 &c, int64_t idx) {
    return ap_mk((idx >= 0 && idx < c.n) ? (float)c.base[idx] : 0.0f,
                 (idx + 1 >= 0 && idx + 1 < c.n) ? (float)c.base[idx + 1] : 0.0f);
}
// output rows as a bounds-checked buffer: stores outside [0, bytes) are dropped
struct ApOutBuf { char *base; int64_t bytes; };
AP_DEV ApOutBuf ap_outbuf_make(void *base, int64_t bytes) { ApOutBuf o; o.base = (char *)base; o.bytes = bytes; return o; }
AP_DEV ap_float2 ap_outbuf_load2(const ApOutBuf &o, unsigned lane_bytes, unsigned uniform_bytes) {
    const uint64_t off = (uint64_t)lane_bytes + uniform_bytes;
    if (lane_bytes < 0x80000000u && off + 8 <= (uint64_t)o.bytes) return *reinterpret_cast<const ap_float2 *>(o.base + off);
    return ap_mk(0.0f, 0.0f);
}
AP_DEV void ap_outbuf_store2(const ApOutBuf &o, unsigned lane_bytes, unsigned uniform_bytes, ap_float2 v) {
    const uint64_t off = (uint64_t)lane_bytes + uniform_bytes;
    if (lane_bytes < 0x80000000u && off + 8 <= (uint64_t)o.bytes) *reinterpret_cast<ap_float2 *>(o.base + off) = v;
}
#else
#define AP_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   // tell the compiler x is wave-uniform
// one clip as a raw buffer resource: the hardware range check returns 0 for every sample
// outside [0, L) — constant padding (stft.py:441-442) with no branch and 32-bit offsets
typedef __amdgpu_buffer_rsrc_t ApClip;
AP_DEV ApClip ap_clip_make(const float *base, int64_t n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)(n * 4), 0x00020000);
}
AP_DEV float ap_clip_load(ApClip c, int64_t idx) {
    // a negative index wraps to a huge unsigned byte offset: out of range -> 0.  The offset is made opaque:
    // left alone the compiler splits it into a register part and an instruction immediate, and a 4-byte
    // load whose register part is negative and whose sum is 0 or 4 comes back as 0 (measured: the first
    // two samples of a clip were lost; the 8-byte loads are not affected)
    int off = (int)(idx * 4);
    AP_PIN(off);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c, off, 0, 0));
}
// samples idx, idx + 1 as one 8-byte load.  idx must be even or the clip start must not fall inside a
// pair: the pair (-1, 0) has a wrapped byte offset and reads as (0, 0), losing sample 0 (measured; the
// launch code only takes this path with even hop and padding).  Past the end the check is per dword.
AP_DEV ap_float2 ap_clip_load2(ApClip c, int idx) {
    typedef int ap_i2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(ap_float2, __builtin_bit_cast(ap_i2, __builtin_amdgcn_raw_buffer_load_b64(c, idx * 4, 0, 0)));
}
// output rows as a raw buffer resource (< 4 GiB): the address of a store is base + lane offset (VGPR) +
// uniform offset (SGPR), so a run of stores to different rows costs no per-store address arithmetic, and
// a lane parks itself by holding an out-of-range offset (the hardware drops the store)
typedef __amdgpu_buffer_rsrc_t ApOutBuf;
AP_DEV ApOutBuf ap_outbuf_make(void *base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
AP_DEV ap_float2 ap_outbuf_load2(ApOutBuf o, unsigned lane_bytes, unsigned uniform_bytes) {
    typedef int ap_i2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(ap_float2, __builtin_amdgcn_raw_buffer_load_b64(o, (int)lane_bytes, (int)uniform_bytes, 0));
}
AP_DEV void ap_outbuf_store2(ApOutBuf o, unsigned lane_bytes, unsigned uniform_bytes, ap_float2 v) {
    typedef int ap_i2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ap_i2, v), o, (int)lane_bytes, (int)uniform_bytes, 0);
}
// 16-bit PCM clip (n even, idx even: the launch code guarantees both): one dword = samples idx, idx + 1,
// converted to float (unscaled: the 1 / 32768 rides on the window); the hardware range check zero-pads
typedef __amdgpu_buffer_rsrc_t ApClip16;
AP_DEV ApClip16 ap_clip16_make(const int16_t *base, int64_t n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t *>(base), 0, (int)(n * 2), 0x00020000);
}
AP_DEV ap_float2 ap_clip16_load2(ApClip16 c, int idx) {
    int off = idx * 2;
    AP_PIN(off);                         // no immediate part: see ap_clip_load
    const int d = __builtin_amdgcn_raw_buffer_load_b32(c, off, 0, 0);
    return ap_mk((float)(short)(d & 0xFFFF), (float)(d >> 16));
}
#endif


// per-lane constants of the 1024-point wave transform
struct ApwLane {
    int lane, qa, qd, k1p;
    float s1, s2;
    ap_float2 rotw;                // (1, 0), or (0, 1) = -i on the quad's lane 3
    const ap_float2 *tw2row;
    ap_float2 tws0h;               // W_2048^lane / 2; bins k = lane + 64 r add W_32^r
    ap_float2 half, halfc;         // (1/2, 1/2), (1/2, -1/2)
};

AP_DEV ApwLane apw_lane_init(int lane, const ap_float2 *TW2, const ap_float2 *tw) {
    ApwLane c;
    c.lane = lane;
    c.qa = lane & 3;                              // position in the quad
    c.s1 = c.qa < 2 ? 1.0f : -1.0f;               // radix-4 stage-1 sign
    c.s2 = (c.qa & 1) ? -1.0f : 1.0f;             // radix-4 stage-2 sign
    c.rotw = c.qa == 3 ? ap_mk(0.0f, 1.0f) : ap_mk(1.0f, 0.0f);   // lane 3 multiplies by -i between stages
    c.qd = ((c.qa & 1) << 1) | (c.qa >> 1);       // output digit held by this lane
    c.k1p = lane >> 2;
    c.tw2row = TW2 + c.qa * 17;
    c.tws0h = ap_scale(tw[lane], 0.5f);
    c.half = ap_mk(0.5f, 0.5f);
    c.halfc = ap_mk(0.5f, -0.5f);
    return c;
}

// workgroup tables: W_64^(a*c) [4][17], W_1024^(lane*k1) [16][64], window as 1024 float pairs
AP_DEV void apw_fill_tables(ap_float2 *tw2, ap_float2 *tw1, ap_float2 *win, const ap_float2 *tw,
                            const float *window, int tid, int nt) {
    // W_64^(a c) with the quad stage's per-lane signs s1 s2 of row a folded in (apm_quad8 works on (s1 s2) v)
    if (tid < 64) {
        const int a = tid >> 4;
        tw2[a * 17 + (tid & 15)] = ap_scale(tw[32 * a * (tid & 15)], (a == 1 || a == 2) ? -1.0f : 1.0f);
    }
    for (int i = tid; i < 16 * 64; i += nt) tw1[i] = tw[2 * (i & 63) * (i >> 6)];
    for (int i = tid; i < APW_NC; i += nt) win[i] = reinterpret_cast<const ap_float2 *>(window)[i];
}

// per-lane constants of the transform (on top of ApwLane)
struct ApmLane {
    float c1, c2;        // stage coefficients of the quad radix-4: -s1, -s2
    float sg;            // s1 s2: folded into the W_64^(a c) twiddles (and into v[0])
    bool rot;            // lane 3 of the quad: multiply by -i between the stages
};

AP_DEV ApmLane apm_lane_init(int lane) {
    ApmLane m;
    const int qa = lane & 3;
    const float s1 = qa < 2 ? 1.0f : -1.0f, s2 = (qa & 1) ? -1.0f : 1.0f;
    m.c1 = -s1;
    m.c2 = -s2;
    m.sg = s1 * s2;
    m.rot = qa == 3;
    return m;
}

// Radix-4 across the quad on values held as h = (s1 s2) v, 8 complex values per call:
//   stage 1: r = h - s1 h[lane ^ 2]         (= s2 (s1 v + v[lane ^ 2]))
//   lane 3:  r *= -i                        (x, y) -> (y, -x)
//   stage 2: out = r - s2 r[lane ^ 1]       (= s2 r' + r'[lane ^ 1] for the unsigned r')
// outputs in bit-reversed lanes as in apw_forward.
#ifdef AP_HOST_EMU
AP_DEV void apm_quad8(ap_float2 *v, const ApmLane &m) {
    for (int i = 0; i < 8; ++i) {
        v[i].x = v[i].x + m.c1 * ap_quad_xor2(v[i].x);
        v[i].y = v[i].y + m.c1 * ap_quad_xor2(v[i].y);
    }
    for (int i = 0; i < 8; ++i) {
        const float rx = v[i].x, ry = v[i].y;
        v[i].x = m.rot ? ry : rx;
        v[i].y = m.rot ? -rx : ry;
    }
    for (int i = 0; i < 8; ++i) {
        v[i].x = v[i].x + m.c2 * ap_quad_xor1(v[i].x);
        v[i].y = v[i].y + m.c2 * ap_quad_xor1(v[i].y);
    }
}
#else
// One asm block: v_fmac_f32_dpp takes the quad-permuted operand straight into the FMA
// (d += c * d[lane ^ 2]); the compiler neither folds a DPP move into a VOP2 FMA here nor knows
// about the DPP read inside an asm, so the block orders its instructions itself: `s_nop 1` covers
// the 2 wait states between an outside VALU write and the first DPP read of that register, and
// inside every DPP read sits >= 14 instructions behind the write of its register.  The halves of
// the complex register pairs are named directly (v[i].x / v[i].y): no unpacking moves.
// lane 3 of the quad: (x, y) -> (y, -x), written to a fresh register pair (operands nx, ny) that stage 2
// carries on with: no copy of the old x, and the pair stays a pair for the packed code that follows.
#define APM_ROT(ix, iy, nx, ny)                            \
    "v_cndmask_b32 %" #nx ", %" #ix ", %" #iy ", %34\n\t"  \
    "v_cndmask_b32 %" #ny ", %" #iy ", -%" #ix ", %34\n\t"
#define APM_S2N(i) "v_fmac_f32_dpp %" #i ", %" #i ", %33 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define APM_S1N(i) "v_fmac_f32_dpp %" #i ", %" #i ", %32 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
AP_DEV void apm_quad8(ap_float2 *v, const ApmLane &m) {
    ap_float2 n[8];
    const unsigned long long rot_mask = 0x8888888888888888ull;      // lane 3 of every quad
    asm volatile(
        "s_nop 1\n\t"
        APM_S1N(0) APM_S1N(1) APM_S1N(2) APM_S1N(3) APM_S1N(4) APM_S1N(5) APM_S1N(6) APM_S1N(7)
        APM_S1N(8) APM_S1N(9) APM_S1N(10) APM_S1N(11) APM_S1N(12) APM_S1N(13) APM_S1N(14) APM_S1N(15)
        APM_ROT(0, 1, 16, 17) APM_ROT(2, 3, 18, 19) APM_ROT(4, 5, 20, 21) APM_ROT(6, 7, 22, 23)
        APM_ROT(8, 9, 24, 25) APM_ROT(10, 11, 26, 27) APM_ROT(12, 13, 28, 29) APM_ROT(14, 15, 30, 31)
        APM_S2N(16) APM_S2N(17) APM_S2N(18) APM_S2N(19) APM_S2N(20) APM_S2N(21) APM_S2N(22) APM_S2N(23)
        APM_S2N(24) APM_S2N(25) APM_S2N(26) APM_S2N(27) APM_S2N(28) APM_S2N(29) APM_S2N(30) APM_S2N(31)
        : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y),
          "+v"(v[4].x), "+v"(v[4].y), "+v"(v[5].x), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[6].y), "+v"(v[7].x), "+v"(v[7].y),
          "=&v"(n[0].x), "=&v"(n[0].y), "=&v"(n[1].x), "=&v"(n[1].y), "=&v"(n[2].x), "=&v"(n[2].y), "=&v"(n[3].x), "=&v"(n[3].y),
          "=&v"(n[4].x), "=&v"(n[4].y), "=&v"(n[5].x), "=&v"(n[5].y), "=&v"(n[6].x), "=&v"(n[6].y), "=&v"(n[7].x), "=&v"(n[7].y)
        : "v"(m.c1), "v"(m.c2), "s"(rot_mask));
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = n[i];
}
#undef APM_S1N
#undef APM_S2N
#undef APM_ROT
#endif

AP_DEV void apm_quad_radix4(ap_float2 (&v)[16], const ApmLane &m) {
    apm_quad8(&v[0], m);
    apm_quad8(&v[8], m);
}

// windowed samples v[j] = z[lane + 64 j]  ->  Z[k] in natural order in the wave's X buffer
// (slots apw_zidx(k)).  16 x 16 x 4: two in-register radix-16 passes around LDS transpose #1,
// the radix-4 across the quad with DPP, then LDS transpose #2.
// The radix-16 butterfly with a scheduling fence after each of its eight radix-4 groups: hipcc's
// scheduler otherwise interleaves all of them (about 80 live temporaries at 2 waves per SIMD), which
// kernels that keep other state in registers across the transform (kernels_stft16.h) cannot afford.
AP_DEV void apw_butterfly16_tight(ap_float2 (&v)[16]) {
    ap_float2 a[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a[r][0] = v[r]; a[r][1] = v[r + 4]; a[r][2] = v[r + 8]; a[r][3] = v[r + 12];
        ap_fft4(a[r][0], a[r][1], a[r][2], a[r][3]);
        AP_SCHED_FENCE();
    }
    const float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
    const float H = 0.70710678118654752440f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {            // twiddles W16^(r p) as in ApButterfly<16>::finish, column by column
        if (p == 1) {
            a[1][1] = ap_mul_fw_c(a[1][1], C1, S1);
            a[2][1] = ap_mul_fw_c(a[2][1], H, H);
            a[3][1] = ap_mul_fw_c(a[3][1], S1, C1);
        } else if (p == 2) {
            a[1][2] = ap_mul_fw_c(a[1][2], H, H);
            a[3][2] = ap_mul_fw_c(a[3][2], -H, H);
        } else if (p == 3) {
            a[1][3] = ap_mul_fw_c(a[1][3], S1, C1);
            a[2][3] = ap_mul_fw_c(a[2][3], -H, H);
            a[3][3] = ap_mul_fw_c(a[3][3], -C1, -S1);
        }
        if (p == 2) ap_fft4_a2mi(a[0][p], a[1][p], a[2][p], a[3][p]);
        else ap_fft4(a[0][p], a[1][p], a[2][p], a[3][p]);
        v[p] = a[0][p]; v[p + 4] = a[1][p]; v[p + 8] = a[2][p]; v[p + 12] = a[3][p];
        AP_SCHED_FENCE();
    }
}

// AP_PRIO(n): issue priority of the wave.  tools/phase_clock.py showed that of a SIMD's two waves the first gets the issue
// slots and the second takes 1.4-1.6 x as long through the same transform, part of it alone on the SIMD with nothing to
// cover its LDS waits.  A kernel that brackets its transform phase with AP_PRIO(3) ... AP_PRIO(0) and passes PRIO = true
// lets a wave lower its priority as it advances, so the wave that is behind is served first (ISTFT: -1.2 ... -1.5 %,
// same box; the STFT measured neutral and leaves it off).
#ifndef AP_HOST_EMU
#define AP_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define AP_PRIO(n) do {} while (0)
#endif

template <bool TO_LDS = true, bool TIGHT = false, bool PRIO = false>
AP_DEV void apw_forward(ap_float2 (&v)[16], ap_float2 *X, const ap_float2 *TW1, const ApwLane &c) {
    const int lane = c.lane;
    if (TIGHT) {                                                   // twiddles fetched after the butterfly
        apw_butterfly16_tight(v);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ap_float2 t1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (8 * h + k) t1[k] = TW1[(8 * h + k) * 64 + lane];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (8 * h + k) v[8 * h + k] = ap_mul_fw(v[8 * h + k], t1[k]);
            AP_SCHED_FENCE();
        }
    } else {
        ap_float2 t1[16];
#pragma unroll
        for (int k = 1; k < 16; ++k) t1[k] = TW1[k * 64 + lane];   // lands during the butterfly
        ApButterfly<16>::run(v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = ap_mul_fw(v[k], t1[k]);
    }
    if (PRIO) AP_PRIO(2);
    {   // transpose #1: (n0 = a + 4b, k1) -> lane (k1, a), register b
        const int a = lane & 3, bq = lane >> 2;
#pragma unroll
        for (int k = 0; k < 16; ++k) X[APW_T1(k * 4 + a) + bq] = v[k];
    }
    AP_WAVE_SYNC();
    ap_float2 t2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = X[APW_T1(lane) + i];
    if (!TIGHT) {
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) t2[cc] = c.tw2row[cc];     // W_64^(a*c)
    }
    AP_WAVE_SYNC();
    const ApmLane m = apm_lane_init(lane);
    if (TIGHT) {
        AP_SCHED_FENCE();
        apw_butterfly16_tight(v);
        v[0] = ap_scale(v[0], m.sg);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < 8; ++k) if (8 * h + k) t2[k] = c.tw2row[8 * h + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (8 * h + k) v[8 * h + k] = ap_mul_fw(v[8 * h + k], t2[k]);
            AP_SCHED_FENCE();
        }
    } else {
        ApButterfly<16>::run(v);
        v[0] = ap_scale(v[0], m.sg);
#pragma unroll
        for (int cc = 1; cc < 16; ++cc) v[cc] = ap_mul_fw(v[cc], t2[cc]);  // the table carries the signs s1 s2
    }
    // radix-4 across the quad (DIF) on v_fmac_f32_dpp, outputs in bit-reversed lanes
    if (PRIO) AP_PRIO(1);
    apm_quad_radix4(v, m);
    // transpose #2: natural order Z[k], k = k1 + 16 c + 256 d (skipped when the caller stores
    // v[cc] = Z[k1p + 16 cc + 256 qd] itself)
    if (TO_LDS) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) X[apw_zidx(c.k1p + 16 * cc + 256 * c.qd)] = v[cc];
        AP_WAVE_SYNC();
    }
}

// paired real split from Z in X: xk[r] = X[lane + 64 r], xm[r] = X[1024 - lane - 64 r] (CONJ = true)
// or its conjugate (CONJ = false, enough for |X|), zh = Z[512] (X[512] = conj Z[512]).  All of Z
// is read before anything else touches X.
//   a = Z[k] + conj Z[1024-k], d = Z[k] - conj Z[1024-k], u = (W^k / 2) d
//   X[k] = a/2 + (-i) u,  conj X[1024-k] = a/2 - (-i) u
template <bool CONJ>
AP_DEV void apw_split(const ap_float2 *X, const ApwLane &c, ap_float2 (&xk)[8], ap_float2 (&xm)[8],
                      ap_float2 &zh) {
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
        // (cos, sin)/2 of angle(lane) + angle(64 r): angle addition with constants
        const ap_float2 w = r == 0 ? c.tws0h : ap_mul_bw_c(c.tws0h, APW_C32(r), APW_S32(r));
        const ap_float2 u = ap_mul_fw(d, w);
        xk[r] = ap_fma_add_mi(a, c.half, u);
        xm[r] = CONJ ? ap_fma_sub_swap(a, c.halfc, u) : ap_fma_sub_mi(a, c.half, u);
    }
}

// PADGEN = 0: constant padding (or none) through the bounds-checked clip buffer;
// PADGEN = 1: edge / reflect padding through the per-sample remap of kernels_generic.h
template <int PMODE, int PADGEN>
__global__ void __launch_bounds__(64 * APW_WAVES, 2) ap_mel2048_wave_kernel(ApMelWaveParams P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APW_X_COMPLEX;
    const ap_float2 *TW2 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw2);   // [4][17]
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);   // [16][64]
    const ap_float2 *WIN = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);   // [1024] pairs
    const ap_float4 *WQ = reinterpret_cast<const ap_float4 *>(ap_smem + P.off_wq);     // [n_quads]
    const ap_int4 *PART = reinterpret_cast<const ap_int4 *>(ap_smem + P.off_parts);    // [n_parts]
    float *partial = reinterpret_cast<float *>(ap_smem + P.off_partial) + wave * P.partial_stride;   // this wave's
    float *otile = reinterpret_cast<float *>(ap_smem + P.off_otile) + wave * P.otile_stride * APW_G;  // [APW_G][stride]
    const int M = P.n_mels;

    // ---------------- workgroup tables in LDS (once; the only workgroup barrier) ------
    {
        const int nt = 64 * APW_WAVES;
        ap_float4 *wq = reinterpret_cast<ap_float4 *>(ap_smem + P.off_wq);
        ap_int4 *part = reinterpret_cast<ap_int4 *>(ap_smem + P.off_parts);
        for (int i = tid; i < P.n_quads; i += nt) wq[i] = reinterpret_cast<const ap_float4 *>(P.quads)[i];
        if (P.n_parts > 64 * APW_PASSES)
            for (int i = tid; i < P.n_parts; i += nt) part[i] = reinterpret_cast<const ap_int4 *>(P.parts)[i];
        apw_fill_tables(reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2),
                        reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1),
                        reinterpret_cast<ap_float2 *>(ap_smem + P.off_win), P.tw, P.window, tid, nt);
    }
    const ApwLane lc = apw_lane_init(lane, TW2, P.tw);
    float *pp = reinterpret_cast<float *>(X);        // |X|^p plane of this wave, aliased on its X buffer
    // frame-invariant contraction state of this lane: its parts and the slot ranges of its rows
    ap_int4 mypart[APW_PASSES];
#pragma unroll
    for (int ps = 0; ps < APW_PASSES; ++ps) {
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

    // every wave is an independent worker: worker w owns the global frames
    // [w*Ftot/W, (w+1)*Ftot/W) of the flattened (clip, frame) stream (equal shares whatever B and
    // T are) and walks them in runs of <= APW_G consecutive frames of one clip.  Adjacent waves own
    // adjacent stretches: their overlapping samples and neighbouring output segments meet in L2.
    const int64_t worker = (int64_t)blockIdx.x * APW_WAVES + wave;
    const int64_t n_workers = (int64_t)gridDim.x * APW_WAVES;
    const int64_t n_frames = P.n_clips * P.T;
    const int64_t f_lo = n_frames * worker / n_workers, f_hi = n_frames * (worker + 1) / n_workers;
    float vmax = -INFINITY;           // running max of this lane's mel values (mfcc's top_db clip needs the global one)
#ifdef AP_DIAG_STAMPS
    unsigned long long ap_dt0, ap_dr0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ap_dt0), "=s"(ap_dr0)::"memory");
#endif
    ap_float2 raw[16];
    auto load_frame = [&](int64_t f) {
        const int64_t b = f / P.T;
        const int64_t t = f - b * P.T;
        const float *yb = P.y + b * P.L;
        const ApClip clip = ap_clip_make(yb, P.L);
        const int64_t base = t * (int64_t)P.hop - P.pad;          // wave-uniform
        // edge / reflect padding only touches the frames that overlap a clip boundary: every
        // other frame takes the same bounds-checked loads as the constant-padding kernel
        const bool inside = !PADGEN || (base >= 0 && base + 2 * APW_NC <= P.L);
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
    // Consecutive frames overlap by n_fft - hop samples, and with hop = 128 s the overlap is a pure
    // register shift: sample pair j of frame t + 1 is pair j + s of frame t.  So only the s new
    // pairs per lane are loaded (4 of 16 at hop 512): a quarter of the load instructions and no
    // re-read of the halo.  Other hops, a new clip and the padded loaders fetch the whole frame.
    auto next_frame = [&](int64_t f) {
        const int64_t b = f / P.T;
        const int64_t t = f - b * P.T;
        if (PADGEN || t == 0 || P.hopj == 0) { load_frame(f); return; }
        const ApClip clip = ap_clip_make(P.y + b * P.L, P.L);
        const int64_t base = t * (int64_t)P.hop - P.pad;
#define APW_SHIFT_LOAD(S)                                                                    \
        {                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 16 - (S); ++j) raw[j] = raw[j + (S)];      \
            _Pragma("unroll") for (int j = 16 - (S); j < 16; ++j) {                           \
                const int64_t p = base + 2 * (lane + 64 * j);                                \
                raw[j] = ap_clip_load2(clip, (int)p);            \
            }                                                                                \
        }
        if (P.hopj == 4) APW_SHIFT_LOAD(4)
        else if (P.hopj == 2) APW_SHIFT_LOAD(2)
        else APW_SHIFT_LOAD(8)
#undef APW_SHIFT_LOAD
    };
    if (f_lo < f_hi) load_frame(f_lo);

    for (int64_t f = f_lo; f < f_hi;) {
        const int64_t b = f / P.T;
        const int64_t t0 = f - b * P.T;
        int64_t run = P.T - t0;
        if (run > APW_G) run = APW_G;
        if (run > f_hi - f) run = f_hi - f;
        const int Gt = (int)run;
        float *ob = P.out + b * (int64_t)M * P.T + t0;

        for (int g = 0; g < Gt; ++g) {
            ap_float2 v[16];
            {
                ap_float2 w[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = WIN[lane + 64 * j];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = ap_mul2(raw[j], w[j]);
            }
            // issue the next frame's loads only AFTER the old samples are consumed: otherwise the
            // compiler hoists them and then has to wait vmcnt(0) for them inside this frame
            AP_SCHED_FENCE();
            if (f + g + 1 < f_hi) next_frame(f + g + 1);            // next frame, in flight during this one
            AP_SCHED_FENCE();
            apw_forward(v, X, TW1, lc);
            {
                ap_float2 xk[8], xm[8], zh;
                apw_split<false>(X, lc, xk, xm, zh);    // all of Z is in registers: the plane overwrites it
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int k = lane + 64 * r;
                    pp[k] = apw_pow2x<PMODE>(xk[r].x, xk[r].y, P.power);
                    pp[APW_NC - k] = apw_pow2x<PMODE>(xm[r].x, xm[r].y, P.power);
                }
                if (lane == 0) pp[APW_NC / 2] = apw_pow2x<PMODE>(zh.x, zh.y, P.power);
            }
            AP_WAVE_SYNC();
            // ---- mel contraction of this frame by its own wave (no workgroup barrier) ----
            // The lane's entry descriptors are frame-invariant (registers): (slot A, group A, slot B,
            // group B).  No branch inside a pass: every lane reads its 4 weight quads (rows 0-1 go
            // with |X|^p groups gA, gA + 1, rows 2-3 with gB, gB + 1; zero where a part ends or the
            // lane is idle) and 4 |X|^p quads, all 8 reads in flight together - with per-group
            // branches the compiler fenced every read with an s_waitcnt.  Half A goes to slot A;
            // half B to slot B (a second short part), or on top of half A (slot B = -1: one long
            // part), or to the dump slot.
#pragma unroll
            for (int ps = 0; ps < APW_PASSES; ++ps) {
                if (64 * ps < P.n_parts) {                         // wave-uniform
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
            for (int p0 = 64 * APW_PASSES; p0 < P.n_parts; p0 += 64) {   // plans with > 256 entries
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
            // a row's partial sums are adjacent; frame t0+g is one column of this wave's
            // (M x APW_G) output tile
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = lane + 64 * i;
                // up to 4 adjacent slots read together (the ones past the row's end are discarded)
                const int cnt = rs1[i] - rs0[i];
                const float p0 = partial[rs0[i]], p1 = partial[rs0[i] + 1], p2 = partial[rs0[i] + 2],
                            p3 = partial[rs0[i] + 3];
                float sum = cnt > 0 ? p0 : 0.0f;
                sum += cnt > 1 ? p1 : 0.0f;
                sum += cnt > 2 ? p2 : 0.0f;
                sum += cnt > 3 ? p3 : 0.0f;
                if (P.max_row_parts > 4)                           // wave-uniform, rare
                    for (int j = rs0[i] + 4; j < rs1[i]; ++j) sum += partial[j];
                if (row < M) {
                    otile[g * P.otile_stride + row] = sum;
                    vmax = fmaxf(vmax, sum);
                }
            }
            for (int row = lane + 128; row < M; row += 64) {       // n_mels > 128
                const int a0 = P.rowstart[row], a1 = P.rowstart[row + 1];
                float sum = 0.0f;
                for (int j = a0; j < a1; ++j) sum += partial[j];
                otile[g * P.otile_stride + row] = sum;
                vmax = fmaxf(vmax, sum);
            }
            AP_WAVE_SYNC();
        }
        // ---- store the tile: APW_G consecutive lanes write one 32-byte row segment of (B,M,T) --
        for (int e = lane; e < M * APW_G; e += 64) {
            const int m = e / APW_G, g = e - m * APW_G;
            if (g < Gt) ob[(int64_t)m * P.T + g] = otile[g * P.otile_stride + m];
        }
        AP_WAVE_SYNC();
        f += Gt;
    }
#ifdef AP_DIAG_STAMPS
    {
        unsigned long long ap_dt1, ap_dr1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ap_dt1), "=s"(ap_dr1)::"memory");
        if (lane == 0 && worker < 256 * 16) { ap_diag_stamps[4 * worker] = ap_dt1 - ap_dt0; ap_diag_stamps[4 * worker + 1] = ap_dr1 - ap_dr0; ap_diag_stamps[4 * worker + 2] = ap_dr0; ap_diag_stamps[4 * worker + 3] = ap_dr1; }
    }
#endif
    if (P.max_key) {                  // one atomic per wave: lanes -> LDS -> lane 0
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
// STFT, n_fft = 2048, complex output (B, 1025, T) with T fastest (stft.py:216).
// The 8 waves of a workgroup transform 8 CONSECUTIVE frames of one clip (one each, same
// register/LDS transform as above), then transpose through LDS (4 chunks of 257 bins,
// double-buffered, one LDS-only workgroup barrier per chunk) so that 8 adjacent lanes write
// 64 contiguous bytes of one bin's row.
//
// The kernel is bound by its stores (the FFT alone runs in a third of the time), and what
// they cost is the number of 64-byte sectors they touch: with T odd the rows of (B, F, T)
// start at arbitrary 8-byte offsets and a segment out[b, k, t0..t0+7] straddles two sectors
// for 7 rows out of 8 (measured: 0.69 ms against 0.38 ms with a sector-aligned row stride).
// So every row is written in sector-ALIGNED windows instead: row k lags by 8 - phi(k) frames,
// phi(k) = frames from t0 to the row's next 64-byte boundary; the thread that owns position
// j of the window keeps the not yet written frame of the previous group in a register
// ("carry", 17 complex per thread) and the workgroup walks a contiguous stretch of one
// clip's groups, so that a window is completed one group later by the same thread.  The
// carries are flushed at the end of a clip / of the workgroup's stretch.
template <int PADGEN>
__global__ void __launch_bounds__(64 * APS_WAVES, 2) ap_stft2048_wave_kernel(ApStftWaveParams P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APW_X_COMPLEX;
    const ap_float2 *TW2 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw2);
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);
    const ap_float2 *WIN = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);
    ap_float2 *OB = reinterpret_cast<ap_float2 *>(ap_smem + P.off_ob);        // [2][257][9]
    apw_fill_tables(reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2),
                    reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1),
                    reinterpret_cast<ap_float2 *>(ap_smem + P.off_win), P.tw, P.window, tid,
                    64 * APS_WAVES);
    const ApwLane lc = apw_lane_init(lane, TW2, P.tw);
    AP_LDS_BARRIER();

    const int F = APW_NC + 1;
    const int sq = tid >> 3, sf = tid & 7;                                   // store role of this thread
    const int T7 = (int)(P.T & 7), Ti = (int)P.T;
    ap_float2 carry[4][2][2], carry_mid = ap_mk(0.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) carry[c][rr][0] = carry[c][rr][1] = ap_mk(0.0f, 0.0f);
    ap_float2 raw[16];
    auto load_frame = [&](int64_t group) {
        const int64_t b = group / P.groups_per_clip;
        const int64_t t = (group - b * P.groups_per_clip) * APS_WAVES + wave;
        const float *yb = P.y + b * P.L;
        const ApClip clip = ap_clip_make(yb, P.L);
        const int64_t base = t * (int64_t)P.hop - P.pad;          // wave-uniform
        // frames beyond T read past the clip: zeros (never stored).  Edge / reflect padding only
        // touches the frames that overlap a clip boundary.
        const bool inside = !PADGEN || t >= P.T || (base >= 0 && base + 2 * APW_NC <= P.L);
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
    // every workgroup owns a contiguous stretch of the (clip, 8-frame group) stream: the 64-byte
    // row segments of consecutive groups are then written back to back by the same CU and meet
    // in its L2 before they reach HBM (a grid-stride order scatters them over the 8 XCDs' L2s and
    // every segment leaves as a partial line)
    const int64_t g_lo = P.n_groups * (int64_t)blockIdx.x / gridDim.x;
    const int64_t g_hi = P.n_groups * ((int64_t)blockIdx.x + 1) / gridDim.x;
    // the edge / reflect instantiation does not prefetch: with both loaders live the 32 extra
    // registers spill inside the store phase, and the kernel is bound by its stores anyway
    if (!PADGEN && g_lo < g_hi) load_frame(g_lo);

    for (int64_t group = g_lo; group < g_hi; ++group) {
        const int64_t b = group / P.groups_per_clip;
        const int64_t t0 = (group - b * P.groups_per_clip) * APS_WAVES;
        ap_float2 *ob = P.out + b * (int64_t)F * P.T + t0;
        if (PADGEN) load_frame(group);
        // (complex index of out[b, 0, t0]) mod 8: the same for every group of a clip
        const int a0 = (int)(((reinterpret_cast<uintptr_t>(P.out) >> 3) + (uint64_t)(b * (int64_t)F * P.T + t0)) & 7);
        const bool have_prev = group > g_lo && t0 > 0;          // carries hold this clip's previous group
        const bool last = group + 1 == g_hi || t0 + APS_WAVES >= P.T;
        const int trem = (int)(P.T - t0);                        // frames t0 + i with i < trem exist

        ap_float2 v[16];
        {
            ap_float2 w[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = WIN[lane + 64 * j];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ap_mul2(raw[j], w[j]);
        }
        AP_SCHED_FENCE();
        if (!PADGEN && group + 1 < g_hi) load_frame(group + 1);              // next group's frame
        AP_SCHED_FENCE();
        ap_float2 xk[8], xm[8], zh;
        apw_forward(v, X, TW1, lc);
        apw_split<true>(X, lc, xk, xm, zh);

        // ---- transposed store: chunk c holds r = 2c, 2c+1 (bins 64r+lane and 1024-64r-lane) ----
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ap_float2 *buf = OB + (c & 1) * (APS_OB_ROWS * APS_OB_ROW);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * c + rr;
                buf[(rr * 128 + lane) * APS_OB_ROW + wave] = xk[r];
                buf[(rr * 128 + 64 + lane) * APS_OB_ROW + wave] = xm[r];
            }
            if (c == 3 && lane == 0) buf[256 * APS_OB_ROW + wave] = ap_mk(zh.x, -zh.y);   // X[512] = conj Z[512]
            AP_LDS_BARRIER();
            // thread (sq = tid / 8, j = tid % 8) owns position j of the aligned windows of rows
            // sq, 64 + sq, 128 + sq, 192 + sq of the chunk: bins 64 r + sq and 1024 - 64 r - sq,
            // r = 2c, 2c + 1.  One LDS read per element: frame (j + phi) mod 8 of this group is
            // either stored now (j >= 8 - phi) or becomes the carry while the old carry is stored.
            int bins[5], slots[5], phi[5];
            ap_float2 x[5];
            const int ne = c == 3 ? 5 : 4;                         // + bin 512 in the last chunk
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * c + (i >> 1);
                bins[i] = (i & 1) ? APW_NC - 64 * r - sq : 64 * r + sq;
                slots[i] = (i >> 1) * 128 + (i & 1) * 64 + sq;
            }
            bins[4] = APW_NC / 2;
            slots[4] = 256;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i < ne) {                                      // all LDS reads of the chunk first
                    phi[i] = (0 - (a0 + bins[i] * T7)) & 7;
                    x[i] = buf[slots[i] * APS_OB_ROW + ((sf + phi[i]) & 7)];
                }
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i < ne) {
                    ap_float2 &cy = i < 4 ? carry[c][i >> 1][i & 1] : carry_mid;
                    const bool mine = i < 4 || tid < APS_WAVES;
                    const bool take = sf < 8 - phi[i];
                    const int dt = phi[i] + sf - 8;                // frame t0 + dt of the row
                    const ap_float2 val = take ? cy : x[i];
                    if (mine && (take ? have_prev : dt < trem)) ob[bins[i] * Ti + dt] = val;
                    if (take) cy = x[i];
                }
            if (last) {                                            // flush: the carries just taken
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < ne) {
                        const bool mine = i < 4 || tid < APS_WAVES;
                        if (mine && sf < 8 - phi[i] && phi[i] + sf < trem) ob[bins[i] * Ti + phi[i] + sf] = x[i];
                    }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// irfft of every frame, n_fft = 2048: S (B, 1025, T) complex -> frames (B, T, 2048)
// (transpose + mx.fft.irfft(n=n_fft), stft.py:292-295).  Mirror image of the STFT kernel:
// the 8 waves of a workgroup take 8 consecutive frames; the (bin, 8 frames) 64-byte
// segments are staged through LDS in 4 double-buffered chunks so every wave ends up with
// its own frame's bins k = lane + 64 r and 1024 - k in registers; Hermitian merge, the
// same 16 x 16 x 4 transform on conjugated data, and 128-byte-coalesced stores straight
// from the quad outputs (no second LDS transpose).
//
// OLA = 1 (ap_istft_f32): the frames never leave the workgroup.  Every wave leaves its windowed
// frame in its own exchange buffer; after one barrier the 512 threads gather the 8 hop output
// positions the group completes (those whose last contributing frame is in the group) — 16-byte
// LDS reads of the <= n_fft/hop frames that cover them, in increasing frame order like
// overlap_add.metal:16-55 — divide by the window-sum-of-squares of the frames that exist, and
// store them; what the group contributes to later positions goes into an LDS carry (two buffers,
// ping-pong) that the next group of the workgroup's contiguous stretch starts from.  A stretch
// that begins inside a clip first runs the preceding group with its stores disabled to rebuild
// the carry; the tail after a clip's last group is emitted from the carry positions.
template <int OLA>
__global__ void __launch_bounds__(64 * APS_WAVES, 2) ap_irfft2048_wave_kernel(ApIrfftWaveParams P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = AP_UNIFORM(tid >> 6);
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APW_X_COMPLEX;
    const ap_float2 *TW2 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw2);
    const ap_float2 *TW1 = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_tw1);
    ap_float2 *IB = reinterpret_cast<ap_float2 *>(ap_smem + P.off_ob);        // [2][257][9]
    {
        ap_float2 *tw2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2);
        ap_float2 *tw1 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw1);
        if (tid < 64)       // signs of the quad stage folded in, as in apw_fill_tables
            tw2[(tid >> 4) * 17 + (tid & 15)] = ap_scale(P.tw[32 * (tid >> 4) * (tid & 15)], ((tid >> 4) == 1 || (tid >> 4) == 2) ? -1.0f : 1.0f);
        for (int i = tid; i < 16 * 64; i += 64 * APS_WAVES) tw1[i] = P.tw[2 * (i & 63) * (i >> 6)];
        if (OLA) {
            float *win = reinterpret_cast<float *>(ap_smem + P.off_win);
            for (int i = tid; i < 2 * APW_NC; i += 64 * APS_WAVES) win[i] = P.window[i];
        }
    }
    const ApwLane lc = apw_lane_init(lane, TW2, P.tw);
    AP_LDS_BARRIER();
    const int F = APW_NC + 1;
    const float scale = 1.0f / 1024.0f;    // 1/n_fft, and the merge above works at half scale

    const int64_t g_lo = P.n_groups * (int64_t)blockIdx.x / gridDim.x;
    const int64_t g_hi = P.n_groups * ((int64_t)blockIdx.x + 1) / gridDim.x;
    // thread (sq = tid / 8, sf = tid % 8) fetches rows sq, 64 + sq, 128 + sq, 192 + sq of every
    // chunk (bins 64 r + sq and 1024 - 64 r - sq, r = 2c, 2c + 1) in SECTOR-ALIGNED windows of 8
    // frames, like the STFT kernel stores them: with T odd a segment S[b, k, t0..t0+7] straddles
    // two 64-byte sectors for 7 rows out of 8 and the kernel fetched 2.7 x its input.  Row k's
    // window g is [t0 + phi - 8, t0 + phi), phi(k) = frames from t0 to the row's next 64-byte
    // boundary; the 8 frames of a group are the tail of window g (carry `wa`, loaded one group
    // earlier) and the head of window g + 1 (`wb`, prefetched under the previous transform), picked
    // with one select per element.
    const int sq = tid >> 3, sf = tid & 7;
    const int Ti = (int)P.T, T7 = (int)(P.T & 7);
    ap_float2 wa[4][4], wb[4][4], wa_mid = ap_mk(0.0f, 0.0f), wb_mid = ap_mk(0.0f, 0.0f);
    auto clip_phase = [&](int64_t b) {      // (complex index of S[b, 0, 0]) mod 8
        return (int)(((reinterpret_cast<uintptr_t>(P.S) >> 3) + (uint64_t)(b * (int64_t)F * P.T)) & 7);
    };
    // the window of every row that starts phi(row) + dt0 frames after frame 0 of clip b
    // (32-bit arithmetic: the launch code bounds T by 2^20, so bin T + t < 2^31)
    auto load_window = [&](ap_float2 (&w)[4][4], ap_float2 &wmid, int64_t b, int dt0) {
        const int a0 = clip_phase(b);
        const ap_float2 *sb = P.S + b * (int64_t)F * P.T;
        const int tb = dt0 + sf;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * c + (i >> 1);
                const int bin = (i & 1) ? APW_NC - 64 * r - sq : 64 * r + sq;
                const int t = tb + ((0 - (a0 + bin * T7)) & 7);
                w[c][i] = (unsigned)t < (unsigned)Ti ? sb[bin * Ti + t] : ap_mk(0.0f, 0.0f);
            }
        if (tid < APS_WAVES) {
            const int t = tb + ((0 - (a0 + (APW_NC / 2) * T7)) & 7);
            wmid = (unsigned)t < (unsigned)Ti ? sb[(APW_NC / 2) * Ti + t] : ap_mk(0.0f, 0.0f);
        }
    };
    // OLA: a stretch that starts inside a clip begins one group early (stores disabled)
    const int64_t g_first = (OLA && g_lo < g_hi && g_lo % P.groups_per_clip != 0) ? g_lo - 1 : g_lo;
    if (g_first < g_hi) {
        const int64_t b = g_first / P.groups_per_clip;
        const int64_t t0 = (g_first - b * P.groups_per_clip) * APS_WAVES;
        load_window(wa, wa_mid, b, (int)t0 - 8);
        load_window(wb, wb_mid, b, (int)t0);
    }
    for (int64_t group = g_first; group < g_hi; ++group) {
        const int64_t b = group / P.groups_per_clip;
        const int64_t t0 = (group - b * P.groups_per_clip) * APS_WAVES;
        const int Gt = (int)((P.T - t0) < APS_WAVES ? (P.T - t0) : APS_WAVES);
        const int a0 = clip_phase(b);                   // t0 is a multiple of 8

        // ---- transpose through LDS: every wave collects its frame's bins in registers -------
        ap_float2 xk[8], xm[8], xh = ap_mk(0.0f, 0.0f);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ap_float2 *buf = IB + (c & 1) * (APS_OB_ROWS * APS_OB_ROW);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 2 * c + (i >> 1);
                const int bin = (i & 1) ? APW_NC - 64 * r - sq : 64 * r + sq;
                const int phi = (0 - (a0 + bin * T7)) & 7;
                // position sf of the windows holds frame (sf + phi) mod 8 of this group
                buf[((i >> 1) * 128 + (i & 1) * 64 + sq) * APS_OB_ROW + ((sf + phi) & 7)] =
                    sf < 8 - phi ? wb[c][i] : wa[c][i];
            }
            if (c == 3 && tid < APS_WAVES) {
                const int phi = (0 - (a0 + (APW_NC / 2) * T7)) & 7;
                buf[256 * APS_OB_ROW + ((sf + phi) & 7)] = sf < 8 - phi ? wb_mid : wa_mid;
            }
            AP_LDS_BARRIER();
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                xk[2 * c + rr] = buf[(rr * 128 + lane) * APS_OB_ROW + wave];
                xm[2 * c + rr] = buf[(rr * 128 + 64 + lane) * APS_OB_ROW + wave];
            }
            if (c == 3) xh = buf[256 * APS_OB_ROW + wave];
        }
        AP_SCHED_FENCE();
        if (group + 1 < g_hi) {
            const int64_t bn = (group + 1) / P.groups_per_clip;
            const int64_t tn = (group + 1 - bn * P.groups_per_clip) * APS_WAVES;
            if (bn == b) {                               // same clip: window g + 1 becomes the carry
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) wa[c][i] = wb[c][i];
                wa_mid = wb_mid;
            } else {
                load_window(wa, wa_mid, bn, (int)tn - 8);
            }
            load_window(wb, wb_mid, bn, (int)tn);
        }
        AP_SCHED_FENCE();
        // ---- Hermitian merge: conj(Z[k]) / 2 and conj(Z[1024-k]) / 2 of the packed inverse -----
        //   a = X[k] + conj X[1024-k], d = X[k] - conj X[1024-k], o = (W^-k / 2) d
        //   conj Z[k] / 2 = conj(a/2 + i o),  conj Z[1024-k] / 2 = a/2 - i o
        ap_float2 v[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            ap_float2 a_k = xk[r], a_m = xm[r];
            if (r == 0 && lane == 0) { a_k.y = 0.0f; a_m.y = 0.0f; }      // DC / Nyquist imaginary parts ignored
            const ap_float2 a = ap_add_conj(a_k, a_m);
            const ap_float2 d = ap_sub_conj(a_k, a_m);
            const ap_float2 w = r == 0 ? lc.tws0h : ap_mul_bw_c(lc.tws0h, APW_C32(r), APW_S32(r));
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
        apw_forward<false>(v, X, TW1, lc);
        // y[n] = conj(Y[n]) / 2048 -> samples 2n, 2n+1;  n = k1p + 16 c + 256 qd
        if (!OLA) {
            if (wave < Gt) {
                float *dst = P.frames + ((b * P.T + t0 + wave) * (int64_t)2048);
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) {
                    const int n = lc.k1p + 16 * cc + 256 * lc.qd;
                    *reinterpret_cast<ap_float2 *>(dst + 2 * n) = ap_mul2(v[cc], ap_mk(scale, -scale));
                }
            }
            continue;
        }
        // ---- fused overlap-add ---------------------------------------------------------------
        const int H = P.hop;
        const int CN = 2 * APW_NC - H;                              // carry length
        const ap_float2 *WINP = reinterpret_cast<const ap_float2 *>(ap_smem + P.off_win);   // (w[2n], w[2n+1])
        const float *WIN = reinterpret_cast<const float *>(ap_smem + P.off_win);
        float *carry_in = reinterpret_cast<float *>(ap_smem + P.off_carry) + (int)(group & 1) * CN;
        float *carry_out = reinterpret_cast<float *>(ap_smem + P.off_carry) + (int)((group + 1) & 1) * CN;
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
        if (t0 == 0)                                                 // a clip starts: nothing carried in
            for (int i = tid; i < CN; i += 64 * APS_WAVES) carry_in[i] = 0.0f;
        AP_LDS_BARRIER();
        const bool emit = group >= g_lo;                             // not the warm-up group
        const bool clip_last = t0 + APS_WAVES >= P.T;
        const float *XF = reinterpret_cast<const float *>(ap_smem);  // frame f at XF + f * 2 APW_X_COMPLEX
        const int n_own = APS_WAVES * H;                             // positions this group completes
        const int hs = H == 256 ? 8 : (H == 512 ? 9 : 10);           // H = 1 << hs
        const int64_t p0 = t0 * (int64_t)H;                          // padded position of r = 0
        const int t0i = (int)t0;
        float *yb = P.y + b * P.out_len;
        const int64_t n0 = p0 - P.out_offset;                        // output index of r = 0
        for (int r = 4 * tid; r < n_own + CN; r += 4 * 64 * APS_WAVES) {
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            if (r < CN) {
                const ap_float4 c4 = *reinterpret_cast<const ap_float4 *>(carry_in + r);
                s0 = c4.x; s1 = c4.y; s2 = c4.z; s3 = c4.w;
            }
            // frames of this group that cover r: f H <= r < f H + n_fft (arithmetic shifts floor)
            const int f_cov = ((r - 2 * APW_NC) >> hs) + 1;          // first covering frame, may be < 0
            const int f_lo = f_cov < 0 ? 0 : f_cov;
            int f_hi = r >> hs;
            if (f_hi > APS_WAVES - 1) f_hi = APS_WAVES - 1;
            for (int f = f_lo; f <= f_hi; ++f) {
                const int sidx = r - (f << hs);                      // sample index in frame f (multiple of 4)
                const ap_float4 q = *reinterpret_cast<const ap_float4 *>(
                    XF + f * (2 * APW_X_COMPLEX) + 2 * apw_zidx(sidx >> 1));
                s0 += q.x; s1 += q.y; s2 += q.z; s3 += q.w;
            }
            if (r >= n_own) {                                        // contributions to the next group's range
                ap_float4 c4; c4.x = s0; c4.y = s1; c4.z = s2; c4.w = s3;
                *reinterpret_cast<ap_float4 *>(carry_out + (r - n_own)) = c4;
            }
            if (emit && (r < n_own || clip_last)) {
                // window-sum-of-squares over the frames of the CLIP that cover the position
                // (relative frame numbers; frames before this group count too)
                int F_lo = f_cov < -t0i ? -t0i : f_cov;
                int F_hi = r >> hs;
                if (F_hi > Ti - 1 - t0i) F_hi = Ti - 1 - t0i;
                float w0 = 0.0f, w1 = 0.0f, w2 = 0.0f, w3 = 0.0f;
                for (int Fi = F_lo; Fi <= F_hi; ++Fi) {
                    const ap_float4 w = *reinterpret_cast<const ap_float4 *>(WIN + (r - Fi * H));
                    w0 += w.x * w.x; w1 += w.y * w.y; w2 += w.z * w.z; w3 += w.w * w.w;
                }
                const int64_t n = n0 + r;
                if (n >= 0 && n + 3 < P.out_len) {
                    yb[n] = s0 / fmaxf(w0, 1e-8f);
                    yb[n + 1] = s1 / fmaxf(w1, 1e-8f);
                    yb[n + 2] = s2 / fmaxf(w2, 1e-8f);
                    yb[n + 3] = s3 / fmaxf(w3, 1e-8f);
                } else {
                    if (n >= 0 && n < P.out_len) yb[n] = s0 / fmaxf(w0, 1e-8f);
                    if (n + 1 >= 0 && n + 1 < P.out_len) yb[n + 1] = s1 / fmaxf(w1, 1e-8f);
                    if (n + 2 >= 0 && n + 2 < P.out_len) yb[n + 2] = s2 / fmaxf(w2, 1e-8f);
                    if (n + 3 >= 0 && n + 3 < P.out_len) yb[n + 3] = s3 / fmaxf(w3, 1e-8f);
                }
            }
        }
        if (emit && clip_last) {                                     // no frame reaches beyond the tail: 0 / 1e-8
            int64_t n = p0 + n_own + CN - P.out_offset;
            if (n < 0) n = 0;
            for (n += tid; n < P.out_len; n += 64 * APS_WAVES) yb[n] = 0.0f;
        }
    }
}
