// Parameter block and geometry constants of the n_fft=2048 wave-per-frame mel kernel
// (kernels_wave.h); HIP-free so the launch code and the CPU emulator can share it.
#pragma once
#include "ap_common.h"

#define APW_NC 1024          // complex points of the packed transform
#define APW_G 8              // consecutive frames one wave processes before it moves to its next tile
#define APW_WAVES 8          // waves per workgroup (2 per SIMD; they only share read-only LDS tables)
// transpose #1 layout: row r (64 rows of 16 complex) starts at complex slot 20 r + 2 (r >> 4):
// conflict-free for both the 16 ds_write_b64 and the 8 ds_read_b128 of a frame (tools/lds_banks.py)
#define APW_T1(r) ((r) * 20 + ((r) >> 4) * 2)
#define APW_X_COMPLEX 1284   // complex slots of one wave's exchange buffer (>= APW_T1(63)+16, >= zidx(1023)+1)
#define APW_PASSES 4          // contraction passes whose part descriptors live in registers (256 parts)
#define APW_TW2_COMPLEX 72   // 4 rows x 17 (padded) of W_64^(a*c), rounded to 16 bytes

struct ApMelWaveParams {
    const float *y;            // (B, L)
    const float *window;       // (2048)
    const ap_float2 *tw;       // (2048) (cos, sin)(2 pi j / 2048)
    const int32_t *parts;      // (n_parts, 4) wave layout: slot, first group, n_groups (0 = idle), first quad
    const float *quads;        // (n_quads, 4) lane-interleaved filter weights: group i at first quad + 64 i
    const int32_t *rowstart;   // (M+1) slot range of every row
    float *out;                // (B, M, T)
    unsigned *max_key;         // NULL, or the order-preserving key of max(out) to raise (one atomic per wave)
    int64_t L, T, tiles_per_clip, n_tiles, n_clips;
    int64_t Ts;                // run kernel: floats between the rows of `out` (T = dense; a multiple of 8 = whole sectors)
    int hop, pad, pad_mode, n_mels, n_parts, n_quads, n_slots;   // n_slots: partial sums per frame
    int hopj;                  // hop / 128 when the next frame reuses this one's registers (2, 4, 8), else 0
    int max_row_parts;         // largest number of parts of one row
    int partial_stride;        // floats between the waves' partial-sum arrays (n_slots + dump slot + 3 read-ahead)
    float power;
    // LDS carve-up (bytes from the start of dynamic LDS)
    int off_tw2, off_tw1, off_win, off_wq, off_parts, off_partial, off_otile, lds_bytes;
    int otile_stride;          // floats between the frames of a wave's [APW_G][n_mels] output tile (= 4 mod 32)
};

// ---- n_fft = 2048 spectral statistics straight from the audio (kernels_mel2048.h) ----------------
struct ApSpecWaveParams {
    const float *y;            // (B, L)
    const float *window;       // (2048)
    const ap_float2 *tw;       // (2048)
    const float *freq;         // (1025) bin centres
    float *centroid, *bandwidth, *rolloff, *flatness;      // (B, T) each, any may be NULL
    int64_t L, T, n_clips;
    int hop, pad, hopj, norm;
    float power, p, roll_percent, amin;
    int off_tw2, off_tw1, off_win, lds_bytes;
};

// ---- n_fft = 2048 STFT (complex output) wave kernel -----------------------------------
#define APS_WAVES 8          // waves per workgroup = frames per group (64-byte row segments of (B,F,T))
#define APS_OB_ROW 9         // complex slots per row of the transpose buffer (8 frames + 1 pad)
#define APS_OB_ROWS 257      // rows per chunk: 2 x (64 bins + 64 mirrored bins) + bin 512

struct ApStftWaveParams {
    const float *y;            // (B, L)
    const float *window;       // (2048)
    const ap_float2 *tw;       // (2048)
    ap_float2 *out;            // (B, 1025, T)
    int64_t L, T, groups_per_clip, n_groups;
    int hop, pad, pad_mode;
    int off_tw2, off_tw1, off_win, off_ob, lds_bytes;
};

// ---- n_fft = 2048 STFT, 16 frames per group (kernels_stft16.h) -------------------------------
// Every wave transforms TWO frames of a 16-frame group; rows of (B, F, T) leave as 128-byte windows.
#define APS16_G 16           // frames per group = complex values per 128-byte row window
#define APS16_OB_ROW 17      // complex slots per row of the transpose buffer (16 frames + 1 pad)
#define APS16_OB_ROWS 129
#define APS16_T2_SPLIT 576         // whole-group tile (T2): rows below live over the waves' exchange buffers, the other 449 beside them    // rows per chunk: 64 bins + 64 mirrored bins (+ bin 512 in the last chunk)

struct ApStft16Params {
    const float *y;            // (B, L)
    const float *window;       // (2048)
    const ap_float2 *tw;       // (2048)
    ap_float2 *out;            // (B, 1025, T) with rows `Ts` complex values apart
    int64_t L, T, Ts, groups_per_clip, n_groups;
    int hop, pad, pad_mode;
    int off_tw2, off_tw1, off_win, off_ob, lds_bytes;
    int stagger;               // start-up delay of workgroup i: ((i / 8) % 4) * stagger * 64 * 127 cycles (see the kernel)
    // fused Griffin-Lim projection (GL = 1 instantiation): previous raw spectrum and the output estimate in the
    // layout of `out` (rows Ts apart), target magnitudes dense (B, 1025, T)
    const ap_float2 *gl_prev;
    const float *gl_mag;
    ap_float2 *gl_rebuilt;
    float gl_momentum;
};

// ---- n_fft = 2048 fused ISTFT, 16-frame loads (kernels_istft16.h) ---------------------------
struct ApIstft16Params {
    const ap_float2 *S;        // (B, 1025, T), rows Ts complex apart
    const ap_float2 *tw;       // (2048)
    const float *window;       // (2048) synthesis window
    float *y;                  // (B, out_len)
    int64_t T, Ts, g16_per_clip, n_g16, out_offset, out_len;
    int hop;                   // 2048 % hop == 0, hop >= 256
    int off_tw2, off_tw1, off_win, off_inv, off_ib, off_carry, lds_bytes;   // off_inv: `hop` reciprocal window sums
};

struct ApIrfftWaveParams {
    const ap_float2 *S;        // (B, 1025, T)
    const ap_float2 *tw;       // (2048)
    float *frames;             // (B, T, 2048)                       [irfft only]
    int64_t T, groups_per_clip, n_groups;
    int off_tw2, off_tw1, off_ob, lds_bytes;
    // fused overlap-add (ap_istft_f32): y[b, i] = sum_t w[s] frame_t[s] / max(sum_t w[s]^2, 1e-8),
    // s = i + out_offset - t hop
    const float *window;       // (2048) synthesis window
    float *y;                  // (B, out_len)
    int64_t out_offset, out_len;
    int hop;                   // 2048 % hop == 0, hop % 4 == 0, hop >= 256
    int off_win, off_carry;    // LDS: window table, two carry buffers of 2048 - hop floats
};
