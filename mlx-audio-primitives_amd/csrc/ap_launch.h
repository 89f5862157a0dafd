// Argument validation and launch geometry for every C-ABI entry point, as plain C++
// shared by the HIP translation unit (audioprims.hip) and the CPU test emulator
// (tests/emu).  Validation mirrors the std::invalid_argument checks of the
// reference extension (overlap_add.cpp:205-222, frame_signal.cpp:128-144,
// pad_signal.cpp:142-147, _frame_impl.py:51-59).
#pragma once
#include <cstdio>

#include <cstdlib>
#include "../../include/audioprims.h"
#include "ap_common.h"
#include "ap_wave_params.h"

char *ap_error_buffer();            // thread-local, 512 bytes (audioprims.hip / emu)

#define AP_FAIL(code, ...)                                         \
    do {                                                           \
        std::snprintf(ap_error_buffer(), 512, __VA_ARGS__);        \
        return (code);                                             \
    } while (0)

static const int64_t kApMaxGrid = 2147483647LL;
static const int64_t kApStreamGrid = 256 * 8;   // memory-bound grid-stride kernels: 8 blocks per CU

static inline int ap_grid_1d(int64_t total, int block, int64_t cap) {
    int64_t g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static inline int ap_prepare_pad(const float *x, int64_t B, int64_t L, int64_t pad, int mode,
                                 const float *out, int *grid) {
    if (!x || !out) AP_FAIL(AP_ERR_INVALID, "pad_signal: NULL buffer");
    if (B <= 0 || L <= 0) AP_FAIL(AP_ERR_INVALID, "pad_signal: signal must be non-empty");
    if (pad < 0) AP_FAIL(AP_ERR_INVALID, "pad_length must be non-negative");
    if (mode < AP_PAD_CONSTANT || mode > AP_PAD_REFLECT)
        AP_FAIL(AP_ERR_INVALID, "Unknown pad mode. Supported: constant, edge, reflect");
    if (mode == AP_PAD_REFLECT && pad > L - 1)
        AP_FAIL(AP_ERR_INVALID, "reflect padding requires pad_length <= signal_length - 1");
    *grid = ap_grid_1d(B * (L + 2 * pad), AP_BLOCK, kApStreamGrid);
    return AP_OK;
}

// The bounds-checked sample loads (ApClip in kernels_wave.h) stand in for constant padding.  A lane reads
// the samples (p, p + 1) with one register offset plus an immediate, and the hardware does not wrap that
// sum: the pair (-1, 0) would read as (0, 0) and lose sample 0.  Such a pair only exists when centred
// frames start at odd sample indices (odd hop or odd padding); those shapes take the index-remapping
// loaders that also serve reflect / edge padding.
template <class PP>
static inline bool ap_clip_loads_ok(const PP &P) {
    return P.pad == 0 || (P.pad_mode == AP_PAD_CONSTANT && !((P.hop | P.pad) & 1));
}

static inline int ap_prepare_frame(const float *x, int64_t B, int64_t L, int frame_length, int hop,
                                   const float *out, int64_t *T, int *grid) {
    if (!x || !out) AP_FAIL(AP_ERR_INVALID, "frame_signal: NULL buffer");
    if (frame_length <= 0) AP_FAIL(AP_ERR_INVALID, "frame_length must be positive");
    if (hop <= 0) AP_FAIL(AP_ERR_INVALID, "hop_length must be positive");
    if (B <= 0 || L < frame_length)
        AP_FAIL(AP_ERR_INVALID, "Signal length (%lld) must be >= frame_length (%d)", (long long)L, frame_length);
    *T = 1 + (L - frame_length) / hop;
    *grid = ap_grid_1d(B * (*T) * frame_length, AP_BLOCK, kApStreamGrid);
    return AP_OK;
}

static inline int ap_prepare_ola(const float *frames, const float *window, int64_t B, int64_t T,
                                 int n_fft, int hop, int64_t out_offset, int64_t out_len,
                                 const float *out, int64_t *blocks_per_row) {
    if (!frames || !window || !out) AP_FAIL(AP_ERR_INVALID, "overlap_add: NULL buffer");
    if (hop <= 0) AP_FAIL(AP_ERR_INVALID, "hop_length must be positive");
    if (out_len <= 0) AP_FAIL(AP_ERR_INVALID, "output_length must be positive");
    if (B <= 0 || T <= 0 || n_fft <= 0)
        AP_FAIL(AP_ERR_INVALID, "frames must have shape (batch, n_frames, n_fft)");
    if (out_offset < 0) AP_FAIL(AP_ERR_INVALID, "out_offset must be non-negative");
    *blocks_per_row = (out_len + AP_BLOCK - 1) / AP_BLOCK;
    if (*blocks_per_row * B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "overlap_add: grid too large");
    return AP_OK;
}

static inline int ap_prepare_resample_poly(const float *x, int64_t B, int64_t L, int up, int down,
                                           const float *taps, int n_taps, int n_pre_remove,
                                           int64_t n_out, const float *out, int64_t *blocks_per_row) {
    if (!x || !taps || !out) AP_FAIL(AP_ERR_INVALID, "resample_poly: NULL buffer");
    if (up <= 0) AP_FAIL(AP_ERR_INVALID, "up must be positive, got %d", up);
    if (down <= 0) AP_FAIL(AP_ERR_INVALID, "down must be positive, got %d", down);
    if (B <= 0 || L <= 0) AP_FAIL(AP_ERR_INVALID, "resample_poly: signal must be non-empty");
    if (n_taps <= 0 || n_pre_remove < 0) AP_FAIL(AP_ERR_INVALID, "resample_poly: bad filter");
    const int64_t expect = (L * up + down - 1) / down;
    if (n_out != expect)
        AP_FAIL(AP_ERR_INVALID, "resample_poly: n_out mismatch (got %lld, expected %lld)",
                (long long)n_out, (long long)expect);
    *blocks_per_row = (n_out + AP_BLOCK - 1) / AP_BLOCK;
    if (*blocks_per_row * B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "resample_poly: grid too large");
    return AP_OK;
}

// LDS-tiled decimator (up == 1): eligibility, output quad-groups per thread and dynamic LDS size.
// Q = 2 / 1 measured 0.74 / 0.75 ms for 1024 x 480 000 -> 160 000 and Q = 4 slower (the bigger span
// leaves 3 workgroups per CU and the staging phase is no longer hidden): start at 2.
static inline bool ap_resample_decim_eligible(int up, int down, int n_taps, int *Q, int *lds_bytes) {
    if (up != 1 || down < 2 || down > 8) return false;
    const int R = 4;
    const int margin = down * (R - 1);
    const int steps = (n_taps + margin + 3) & ~3;
    for (int q = 2; q >= 1; q >>= 1) {
        const int span = AP_BLOCK * R * q * down + steps;
        const int bytes = (steps * R + span) * (int)sizeof(float);
        if (bytes <= 64 * 1024) { *Q = q; *lds_bytes = bytes; return true; }
    }
    return false;
}

// two outputs per thread (ap_resample_decim2_kernel) when their window wastes fewer positions than four outputs' does
static inline bool ap_resample_decim2_eligible(int up, int down, int n_taps, int *Q, int *lds_bytes) {
    if (up != 1 || down < 2 || down > 8) return false;
    const int steps2 = (n_taps + down + 3) & ~3, steps4 = (n_taps + 3 * down + 3) & ~3;
    if (steps2 >= steps4) return false;
    const int q = 4;
    const int span = AP_BLOCK * 2 * q * down + steps2;
    const int bytes = (steps2 * 2 + span) * (int)sizeof(float);
    if (bytes > 64 * 1024) return false;
    *Q = q;
    *lds_bytes = bytes;
    return true;
}

static inline int ap_prepare_stft(ApStftParams &P, const float *y, int64_t B, int64_t L, int n_fft,
                                  int hop, const float *window, const float *tw, int center,
                                  int pad_mode, int64_t T) {
    if (!y || !window || !tw) AP_FAIL(AP_ERR_INVALID, "stft: NULL buffer");
    if (n_fft <= 0) AP_FAIL(AP_ERR_INVALID, "n_fft must be positive, got %d", n_fft);
    if (hop <= 0) AP_FAIL(AP_ERR_INVALID, "hop_length must be positive, got %d", hop);
    if (B <= 0 || L <= 0) AP_FAIL(AP_ERR_INVALID, "stft: signal must be non-empty");
    if (pad_mode < AP_PAD_CONSTANT || pad_mode > AP_PAD_REFLECT)
        AP_FAIL(AP_ERR_INVALID, "Unknown pad_mode. Supported: reflect, constant, edge");
    const int pad = center ? n_fft / 2 : 0;
    if (center && pad_mode == AP_PAD_REFLECT && pad > L - 1)
        AP_FAIL(AP_ERR_INVALID, "reflect padding requires n_fft//2 <= signal_length - 1");
    const int64_t Texp = ap_n_frames(L, n_fft, hop, center);
    if (Texp <= 0)
        AP_FAIL(AP_ERR_INVALID,
                "Signal length (%lld) must be >= frame_length (%d). Consider padding the signal.",
                (long long)(L + 2 * pad), n_fft);
    if (T != Texp)
        AP_FAIL(AP_ERR_INVALID, "stft: n_frames mismatch (got %lld, expected %lld)", (long long)T,
                (long long)Texp);
    if (ap_make_plan(n_fft, &P.plan) != 0)
        AP_FAIL(AP_ERR_UNSUPPORTED, "n_fft=%d: cannot build FFT plan", n_fft);
    if (ap_make_tile(&P.plan, T, &P.tile) != 0)
        AP_FAIL(AP_ERR_UNSUPPORTED, "n_fft=%d does not fit the %d KiB LDS of one CU", n_fft,
                AP_LDS_MAX / 1024);
    P.y = y;
    P.window = window;
    P.tw = reinterpret_cast<const ap_float2 *>(tw);
    P.L = L;
    P.T = T;
    P.tiles_per_clip = (T + P.tile.G - 1) / P.tile.G;
    P.n_clips = B;
    P.hop = hop;
    P.pad = pad;
    P.pad_mode = pad_mode;
    P.n_bins = n_fft / 2 + 1;
    P.out_c = nullptr;
    P.out_mel = nullptr;
    P.fb = nullptr;
    P.band_lo = nullptr;
    P.band_len = nullptr;
    P.n_mels = 0;
    P.power = 2.0f;
    P.parts = nullptr;
    P.quads = nullptr;
    P.rowstart = nullptr;
    P.n_parts = 0;
    P.n_quads = 0;
    if (P.tiles_per_clip * B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "stft: grid too large");
    return AP_OK;
}

static inline int ap_prepare_mel(ApStftParams &P, const float *fb, const int32_t *plan,
                                 const int32_t *desc, int n_mels, float power, float *out) {
    if (!out || !fb) AP_FAIL(AP_ERR_INVALID, "melspectrogram: NULL buffer");
    if (n_mels <= 0) AP_FAIL(AP_ERR_INVALID, "n_mels must be positive, got %d", n_mels);
    if ((plan == nullptr) != (desc == nullptr))
        AP_FAIL(AP_ERR_INVALID, "plan and desc must both be given or both be NULL");
    if (desc && (desc[1] != n_mels || desc[2] != P.n_bins))
        AP_FAIL(AP_ERR_INVALID, "mel plan was built for (%d, %d), call has (%d, %d)", desc[1],
                desc[2], n_mels, P.n_bins);
    P.out_mel = out;
    P.fb = fb;
    const bool banded = plan && (desc[0] & AP_PLAN_BANDED);
    P.band_lo = banded ? plan + desc[4] : nullptr;
    P.band_len = banded ? plan + desc[5] : nullptr;
    P.n_mels = n_mels;
    P.power = power;
    const bool has_parts = plan && (desc[0] & AP_PLAN_PARTS);
    P.parts = has_parts ? plan + desc[6] : nullptr;
    P.n_parts = has_parts ? desc[7] : 0;
    P.quads = has_parts ? reinterpret_cast<const float *>(plan + desc[8]) : nullptr;
    P.n_quads = has_parts ? desc[9] : 0;
    P.rowstart = has_parts ? plan + desc[10] : nullptr;
    return AP_OK;
}

static inline int ap_align16(int x) { return (x + 15) & ~15; }

// n_fft = 2048 wave-per-frame mel kernel (kernels_wave.h): eligibility and geometry
static inline bool ap_mel_wave_eligible(int n_fft, const int32_t *plan, const int32_t *desc) {
    return n_fft == 2048 && plan && desc && (desc[0] & AP_PLAN_PARTS) &&
           !(desc[0] & AP_PLAN_FORCE_GENERIC);
}

static inline int ap_prepare_mel_wave(ApMelWaveParams &W, const ApStftParams &P, int64_t B,
                                      const int32_t *plan, const int32_t *desc, int *grid) {
    if (P.L > (1 << 28)) return 1;                        // 32-bit sample offsets in the bounds-checked loads
    const int n_waves = APW_WAVES, x_complex = APW_X_COMPLEX;
    const int M = P.n_mels;
    W.y = P.y;
    W.window = P.window;
    W.tw = P.tw;
    W.parts = plan + desc[11];
    W.n_parts = desc[12];
    W.quads = reinterpret_cast<const float *>(plan + desc[13]);
    W.n_quads = desc[14];
    W.n_slots = desc[7];
    W.hopj = (P.hop == 256 || P.hop == 512 || P.hop == 1024) ? P.hop / 128 : 0;
    W.max_row_parts = desc[15];
    W.partial_stride = (W.n_slots + 1 + 3 + 3) & ~3;
    if (W.partial_stride < 64) W.partial_stride = 64;          // the max reduction stages 64 lanes there
    W.rowstart = plan + desc[10];
    W.out = P.out_mel;
    W.max_key = nullptr;
    W.L = P.L;
    W.T = P.T;
    W.tiles_per_clip = (P.T + APW_G - 1) / APW_G;
    W.n_tiles = W.tiles_per_clip * B;
    W.n_clips = B;
    W.hop = P.hop;
    W.pad = P.pad;
    W.pad_mode = P.pad_mode;
    W.n_mels = M;
    W.power = P.power;
    int off = n_waves * x_complex * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += APW_TW2_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 16 * 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += APW_NC * (int)sizeof(ap_float2);
    W.off_wq = off; off += ap_align16(W.n_quads * 16);
    // part descriptors beyond the APW_PASSES register-resident passes are read from LDS
    W.off_parts = off; off += W.n_parts > 64 * APW_PASSES ? ap_align16(W.n_parts * 16) : 0;
    W.off_partial = off; off += ap_align16(n_waves * W.partial_stride * 4);
    W.otile_stride = ((M + 27) / 32) * 32 + 4;              // >= M and = 4 (mod 32): conflict-free both ways
    W.off_otile = off; off += ap_align16(n_waves * W.otile_stride * APW_G * 4);
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;     // does not fit: caller falls back to the generic engine
    // persistent: one 8-wave workgroup per CU; every wave strides over the tiles on its own
    int64_t g = (W.n_tiles + n_waves - 1) / n_waves;
    if (g > 256) g = 256;
    *grid = (int)g;
    return AP_OK;
}

// n_fft = 2048 run kernel (kernels_mel2048.h): 12 waves per CU, output runs in registers.  Returns 1
// when the shape is not one it serves (the caller then takes ap_mel2048_wave_kernel).
//   n_waves / x_complex / partial_off: APM_WAVES, APW_X_COMPLEX, APM_PARTIAL_OFF of the kernel header
static inline int ap_prepare_mel_run(ApMelWaveParams &W, const ApStftParams &P, int64_t B, const int32_t *plan,
                                     const int32_t *desc, int n_waves, int x_complex, int partial_off,
                                     int *n_pass, int *grid) {
    // (reflect / edge padding and odd hops: the caller picks the kernel's index-remapping input mode)
    if (P.n_mels > 128 || desc[12] > 256 || desc[15] > 4) return 1;   // (rows of more than 4 parts: the tile kernel; a loop for them
                                                                      // in the run kernel cost the headline shape 6 VGPRs and ~2 %)
    if (P.T > (1 << 24) || P.L > (1 << 28)) return 1;          // 32-bit frame and sample arithmetic in the loop
    W.y = P.y;
    W.window = P.window;
    W.tw = P.tw;
    W.parts = plan + desc[11];
    W.n_parts = desc[12];
    W.quads = reinterpret_cast<const float *>(plan + desc[13]);
    W.n_quads = desc[14];
    W.n_slots = desc[7];
    W.hopj = P.hop == 512 ? 4 : 0;
    W.max_row_parts = desc[15];
    W.partial_stride = 0;
    // partial sums live inside the wave's exchange buffer: slots + dump slot + 3 read-ahead, and the
    // 64 lane maxima of the max reduction
    if (partial_off + W.n_slots + 4 > 2 * x_complex || partial_off + 64 > 2 * x_complex) return 1;
    W.rowstart = plan + desc[10];
    W.out = P.out_mel;
    W.max_key = nullptr;
    W.L = P.L;
    W.T = P.T;
    W.Ts = P.T;                                            // dense rows; ap_melspec_rows_f32 overwrites it
    W.tiles_per_clip = 0;
    W.n_tiles = 0;
    W.n_clips = B;
    W.hop = P.hop;
    W.pad = P.pad;
    W.pad_mode = P.pad_mode;
    W.n_mels = P.n_mels;
    W.power = P.power;
    const int pass = (W.n_parts + 63) / 64;
    if (pass < 1 || pass > 4 || W.n_quads < 256 * pass) return 1;   // a pass reads 4 weight quads per lane
    *n_pass = pass;
    int off = n_waves * x_complex * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += APW_TW2_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 16 * 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += APW_NC * (int)sizeof(ap_float2);
    W.off_wq = off; off += ap_align16(W.n_quads * 16);
    W.off_parts = off;
    W.off_partial = 0;
    W.off_otile = 0;
    W.otile_stride = 0;
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    // persistent: one 12-wave workgroup per CU, every wave a contiguous stretch of >= 8 frames
    const int64_t n_frames = B * P.T;
    int64_t g = (n_frames + (int64_t)n_waves * 8 - 1) / ((int64_t)n_waves * 8);
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    *grid = (int)g;
    return AP_OK;
}

// fused spectral statistics from audio (kernels_mel2048.h, ap_spec2048_run_kernel): n_fft = 2048, sample
// loads that stand in for constant padding.  Returns 1 when the kernel does not apply.
static inline int ap_prepare_spec_run(ApSpecWaveParams &W, const ApStftParams &P, int64_t B, int n_waves,
                                      int x_complex, int *grid) {
    if (P.plan.n != 2048 || !ap_clip_loads_ok(P)) return 1;
    if (P.T > (1 << 24) || P.L > (1 << 28)) return 1;
    W.y = P.y;
    W.window = P.window;
    W.tw = P.tw;
    W.L = P.L;
    W.T = P.T;
    W.n_clips = B;
    W.hop = P.hop;
    W.pad = P.pad;
    W.hopj = P.hop == 512 ? 4 : 0;
    int off = n_waves * x_complex * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += APW_TW2_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 16 * 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += APW_NC * (int)sizeof(ap_float2);
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    const int64_t n_frames = B * P.T;
    int64_t g = (n_frames + (int64_t)n_waves * 8 - 1) / ((int64_t)n_waves * 8);
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    *grid = (int)g;
    return AP_OK;
}

// Eight-frames-per-wave kernels (kernels_frames8.h): n_fft = 16 R, R in {16, 25, 32}.  The geometry
// (block stride, plane floats, weight row floats, window in LDS) comes from ApqGeom<R> at the call site.
// Returns 1 when the kernel does not apply.
struct ApFrames8Geom { int R, bs, plane_floats, wmax_max, win_lds; };
template <class W8>
static inline int ap_prepare_frames8(W8 &W, const ApStftParams &P, int64_t B, bool mel, const int32_t *plan,
                                     const int32_t *desc, int n_waves, const ApFrames8Geom &G, int *grid) {
    if (P.plan.n != 16 * G.R) return 1;
    // (reflect / edge padding, odd hops: the caller launches the PADGEN instantiation)
    if (P.T > (1 << 19) || P.L > (1 << 28)) return 1;      // 32-bit frame, sample and output-offset arithmetic (a clip's STFT rows < 2 GiB)
    if (mel) {
        if (!(plan && desc && (desc[0] & AP_PLAN_BANDED) && (desc[0] & AP_PLAN_PARTS)) || (desc[0] & AP_PLAN_FORCE_GENERIC)) return 1;
        if (P.n_mels > 128 || desc[15] > 4) return 1;      // <= 4 parts of <= 16 bins per filter: bands of <= 64 bins
    }
    W.y = P.y;
    W.window = P.window;
    W.tw = P.tw;
    W.fb = mel ? P.fb : nullptr;
    W.band_lo = mel ? P.band_lo : nullptr;
    W.band_len = mel ? P.band_len : nullptr;
    W.out = mel ? P.out_mel : reinterpret_cast<float *>(P.out_c);
    W.max_key = nullptr;
    W.L = P.L;
    W.T = P.T;
    W.Ts = P.T;                                            // dense rows; the padded-row entry points overwrite it
    W.plain_stores = 1;
    W.n_clips = B;
    W.groups_per_clip = (P.T + 7) / 8;
    W.n_groups = W.groups_per_clip * B;
    W.hop = P.hop;
    W.pad = P.pad;
    W.pad_mode = P.pad_mode;
    W.n_mels = mel ? P.n_mels : 0;
    W.wmax = (mel && desc[15] > 2) ? G.wmax_max : G.wmax_max - 32;
    W.power = P.power;
    const int m8 = mel ? 8 * ((P.n_mels + 7) / 8) : 0;
    int off = 0;
    W.off_t = off; off += 8 * G.bs * (int)sizeof(ap_float2);
    W.off_s = off; off += 8 * G.bs * (int)sizeof(ap_float2);
    W.off_win = off; off += G.win_lds ? 8 * G.R * (int)sizeof(ap_float2) : 0;
    W.off_w = off; off += m8 * W.wmax * 4;
    W.off_lo = off; off += ap_align16((m8 + m8 / 8) * 4);
    W.off_plane = off; off += mel ? n_waves * 8 * G.plane_floats * 4 : 0;
    // mel: a [m8][8] output tile per wave so that a group's rows leave as 16-byte stores (off_stage = 0: every lane
    // stores its own values); dropped when it does not fit
    W.off_stage = 0;
    if (mel && off + n_waves * m8 * 8 * 4 <= AP_LDS_MAX) { W.off_stage = off; off += n_waves * m8 * 8 * 4; }
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    int64_t g = (W.n_groups + (int64_t)n_waves * 2 - 1) / ((int64_t)n_waves * 2);    // >= 2 groups per wave
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    *grid = (int)g;
    return AP_OK;
}

// inverse eight-frames-per-wave kernel (kernels_frames8.h, ap_irfft8_wave_kernel): frames from the spectrum
template <class W8>
static inline int ap_prepare_irfft8(W8 &W, const ApIrfftParams &P, int64_t B, int R, int bs, int n_waves, int *grid) {
    if (P.plan.n != 16 * R) return 1;
    if (P.T > (1 << 19)) return 1;                          // a clip's spectrum rows < 2 GiB (32-bit offsets)
    W.S = reinterpret_cast<const ap_float2 *>(P.S);
    W.tw = P.tw;
    W.frames = P.frames;
    W.T = P.T;
    W.n_clips = B;
    W.groups_per_clip = (P.T + 7) / 8;
    W.n_groups = W.groups_per_clip * B;
    int off = 0;
    W.off_t = off; off += 8 * bs * (int)sizeof(ap_float2);
    W.off_s = off; off += 8 * bs * (int)sizeof(ap_float2);
    W.lds_bytes = off;
    int64_t g = (W.n_groups + (int64_t)n_waves * 2 - 1) / ((int64_t)n_waves * 2);
    if (g > 1024) g = 1024;                                  // 8 KB of LDS, 8 waves: up to four workgroups' worth of work per CU
    if (g < 1) g = 1;
    *grid = (int)g;
    return AP_OK;
}

// fused ISTFT of the eight-frames-per-wave family (kernels_frames8.h, ap_istft8_wave_kernel): n_fft 512 / 400 /
// 256 with n_fft / 8 <= hop <= n_fft.  `ap_istft8_lds_bytes` < 0: the shape is not served.
static inline int ap_istft8_block_stride(int n_fft) {       // ApqGeom<R>::BS
    const int R = n_fft / 16;
    return R + (R % 2 == 0 ? 1 : 0);
}
static inline int ap_istft8_lds_bytes(int64_t T, int n_fft, int hop, int n_waves) {
    if (n_fft != 512 && n_fft != 400 && n_fft != 256) return -1;
    if (hop <= 0 || hop > n_fft || 8 * hop < n_fft) return -1;
    if (T <= 0 || T > (1 << 19)) return -1;
    const int span = 7 * hop + n_fft;
    const int64_t bytes = 2 * 8 * ap_istft8_block_stride(n_fft) * 8 + n_fft * 4 + ap_align16(hop * 4) + (int64_t)n_waves * span * 4;
    return bytes <= AP_LDS_MAX ? (int)bytes : -1;
}
template <class W8>
static inline int ap_prepare_istft8(W8 &W, const float *S, const float *tw, int64_t B, int64_t T, int n_fft,
                                    const float *window, int hop, int64_t out_offset, int64_t out_len, float *y,
                                    int n_waves, int *grid) {
    if (ap_istft8_lds_bytes(T, n_fft, hop, n_waves) < 0 || out_offset < 0) return 1;
    W.S = reinterpret_cast<const ap_float2 *>(S);
    W.tw = reinterpret_cast<const ap_float2 *>(tw);
    W.window = window;
    W.y = y;
    W.T = T;
    W.Ts = T;                                              // dense rows; ap_istft_rows_f32 overwrites it
    W.n_clips = B;
    W.groups_per_clip = (T + 7) / 8;
    W.n_groups = W.groups_per_clip * B;
    W.out_offset = out_offset;
    W.out_len = out_len;
    W.hop = hop;
    W.span = 7 * hop + n_fft;
    const int bs = ap_istft8_block_stride(n_fft);
    int off = 0;
    W.off_t = off; off += 8 * bs * 8;
    W.off_s = off; off += 8 * bs * 8;
    W.off_w2 = off; off += n_fft * 4;
    W.off_wss = off; off += ap_align16(hop * 4);
    W.off_acc = off; off += n_waves * W.span * 4;
    W.lds_bytes = off;
    // stretches of >= 4 groups: every stretch that starts inside a clip pays one warm-up group
    int64_t g = (W.n_groups + (int64_t)n_waves * 4 - 1) / ((int64_t)n_waves * 4);
    if (g > 512) g = 512;
    if (g < 1) g = 1;
    *grid = (int)g;
    return AP_OK;
}

// n_fft = 1024 wave-per-frame mel kernel (kernels_wave512.h): constant padding, plan with parts,
// at most 128 filters (two rows per lane).  Returns 1 when it does not apply.
struct ApMelWave512Params;
template <class W512>
static inline int ap_prepare_mel_wave512(W512 &W, const ApStftParams &P, int64_t B, const int32_t *plan,
                                         const int32_t *desc, int n_waves, int x_complex, int passes_reg,
                                         int *grid) {
    if (!(plan && desc && (desc[0] & AP_PLAN_PARTS)) || (desc[0] & AP_PLAN_FORCE_GENERIC)) return 1;
    // (reflect / edge padding, odd hops: the caller launches the PADGEN instantiation)
    if (P.T > (1 << 24) || P.L > (1 << 28)) return 1;          // 32-bit frame and sample arithmetic in the loop
    if (P.n_mels > 128) return 1;
    W.y = P.y;
    W.window = P.window;
    W.tw = P.tw;
    W.parts = plan + desc[11];
    W.n_parts = desc[12];
    W.quads = reinterpret_cast<const float *>(plan + desc[13]);
    W.n_quads = desc[14];
    W.n_slots = desc[7];
    W.max_row_parts = desc[15];
    W.partial_stride = (W.n_slots + 1 + 3 + 3) & ~3;
    if (W.partial_stride < 64) W.partial_stride = 64;          // the max reduction stages 64 lanes there
    W.rowstart = plan + desc[10];
    W.out = P.out_mel;
    W.max_key = nullptr;
    W.L = P.L;
    W.Ts = P.T;                                                // dense rows; ap_melspec_rows_f32 overwrites it
    W.T = P.T;
    W.n_clips = B;
    W.hop = P.hop;
    W.pad = P.pad;
    W.pad_mode = P.pad_mode;
    W.n_mels = P.n_mels;
    W.power = P.power;
    W.hopj = (P.hop == 128 || P.hop == 256 || P.hop == 512) ? P.hop / 128 : 0;
    int off = n_waves * x_complex * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 8 * 64 * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += 512 * (int)sizeof(ap_float2);
    W.off_wq = off; off += ap_align16(W.n_quads * 16);
    W.off_parts = off; off += W.n_parts > 64 * passes_reg ? ap_align16(W.n_parts * 16) : 0;
    W.off_partial = off; off += ap_align16(n_waves * W.partial_stride * 4);
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    const int64_t n_frames = B * P.T;
    int64_t g = (n_frames + (int64_t)n_waves * 8 - 1) / ((int64_t)n_waves * 8);   // >= 8 frames per wave
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    *grid = (int)g;
    return AP_OK;
}

// n_fft = 1024 STFT wave kernel geometry (kernels_wave512.h); returns 1 when it does not apply
template <class W512>
static inline int ap_prepare_stft_wave512(W512 &W, const ApStftParams &P, int64_t B, int n_waves,
                                          int x_complex, int ob_complex, int *grid) {
    if (P.L > (1 << 28)) return 1;
    if (P.T > (1 << 20)) return 1;                        // 32-bit row offsets in the store phase
    W.y = P.y;
    W.window = P.window;
    W.tw = P.tw;
    W.out = P.out_c;
    W.L = P.L;
    W.T = P.T;
    W.groups_per_clip = (P.T + n_waves - 1) / n_waves;
    W.n_groups = W.groups_per_clip * B;
    W.hop = P.hop;
    W.pad_mode = P.pad_mode;
    W.padgen = ap_clip_loads_ok(P) ? 0 : 1;
    W.pad = P.pad;
    int off = n_waves * x_complex * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 8 * 64 * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += 512 * (int)sizeof(ap_float2);
    W.off_ob = off; off += ap_align16(ob_complex * (int)sizeof(ap_float2));
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    int64_t g = W.n_groups < 512 ? W.n_groups : 512;      // persistent: two workgroups per CU
    *grid = (int)g;
    return AP_OK;
}

// fused n_fft = 1024 ISTFT (kernels_wave512.h): returns 1 when it does not apply
static inline bool ap_istft1024_fused_shape(int64_t B, int64_t T, int n_fft, int hop, int64_t out_offset) {
    if (n_fft != 1024 || B <= 0 || T <= 0 || T > (1 << 20)) return false;
    if (hop != 128 && hop != 256 && hop != 512) return false;
    if (out_offset % 4 != 0) return false;
    return ((T + 7) / 8) * B >= 64;
}
template <class W512>
static inline int ap_prepare_istft_wave512(W512 &W, const float *S, const float *tw, int64_t B, int64_t T,
                                           const float *window, int hop, int64_t out_offset, int64_t out_len,
                                           float *y, int n_waves, int x_complex, int ib_complex, int *grid) {
    if (!ap_istft1024_fused_shape(B, T, 1024, hop, out_offset)) return 1;
    W.S = reinterpret_cast<const ap_float2 *>(S);
    W.tw = reinterpret_cast<const ap_float2 *>(tw);
    W.window = window;
    W.y = y;
    W.T = T;
    W.groups_per_clip = (T + n_waves - 1) / n_waves;
    W.n_groups = W.groups_per_clip * B;
    W.out_offset = out_offset;
    W.out_len = out_len;
    W.hop = hop;
    int off = n_waves * x_complex * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 8 * 64 * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += 1024 * (int)sizeof(float);
    W.off_ib = off; off += ap_align16(ib_complex * (int)sizeof(ap_float2));
    W.off_carry = off; off += 2 * (1024 - hop) * (int)sizeof(float);
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    int64_t g = W.n_groups / 4;                            // >= 4 groups per stretch
    if (g > 512) g = 512;                                  // two workgroups per CU
    if (g < 1) g = 1;
    *grid = (int)g;
    return AP_OK;
}

// compile-time specialised engine (kernels_ct.h): tile height and LDS bytes for complex length nc
// LDS geometry of the compile-time engine (shared with kernels_ct.h).  The first pass (radix R0)
// leaves its output transposed with the odd row stride PQ = (nc / R0) | 1, so a frame needs
// max(R0 PQ, nc + 1) complex slots; the frame stride FS and the |X|^p plane stride PS are = 4
// (mod 32) so that the 8 frames x 4 bins a half-wave touches in the split / power / contraction
// steps fall into different banks.
static inline int ap_ct_round4mod32(int v) { return ((v + 27) / 32) * 32 + 4; }
static inline int ap_ct_fs(int n_fft) {
    const int nc = n_fft / 2, r0 = n_fft == 400 ? 8 : 16;
    const int pq = (nc / r0) | 1;
    return ap_ct_round4mod32(r0 * pq > nc + 1 ? r0 * pq : nc + 1);
}
static inline int ap_ct_ps(int n_fft) { return ap_ct_round4mod32(n_fft / 2 + 1 + 3); }

static inline bool ap_ct_config(int n_fft, int n_parts, int n_quads, int n_mels, int *G, int *lds_bytes) {
    int g;
    if (n_fft == 400) g = 8;
    else if (n_fft == 512) g = 8;
    else if (n_fft == 1024) g = 8;
    else return false;
    *G = g;
    int bytes = (2 * g * ap_ct_fs(n_fft) + n_fft) * (int)sizeof(ap_float2);
    // mel plan tables + partial sums [n_parts][G] + rowstart
    bytes += n_quads * 16 + n_parts * 16 + n_parts * g * 4 + (n_parts > 0 ? (n_mels + 1) * 4 : 0) + 64;
    *lds_bytes = bytes;
    return bytes <= AP_LDS_MAX;
}

// n_fft = 2048 STFT wave kernel geometry
static inline int ap_prepare_stft_wave(ApStftWaveParams &W, const ApStftParams &P, int64_t B, int *grid) {
    if (P.L > (1 << 28)) return 1;                        // 32-bit sample offsets in the bounds-checked loads
    W.y = P.y;
    W.window = P.window;
    W.tw = P.tw;
    W.out = P.out_c;
    W.L = P.L;
    W.T = P.T;
    W.groups_per_clip = (P.T + APS_WAVES - 1) / APS_WAVES;
    W.n_groups = W.groups_per_clip * B;
    W.hop = P.hop;
    W.pad = P.pad;
    W.pad_mode = P.pad_mode;
    int off = APS_WAVES * APW_X_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += APW_TW2_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 16 * 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += APW_NC * (int)sizeof(ap_float2);
    W.off_ob = off; off += ap_align16(2 * APS_OB_ROWS * APS_OB_ROW * (int)sizeof(ap_float2));
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    if (P.T > (1 << 20)) return 1;                        // 32-bit row offsets (1024 T complex) in the store phase
    int64_t g = W.n_groups < 256 ? W.n_groups : 256;      // persistent: one workgroup per CU
    *grid = (int)g;
    return AP_OK;
}

// n_fft = 2048 STFT, 16 frames per group (kernels_stft16.h).  Ts = complex values between the rows of
// `out` (T for the reference's contiguous layout); *aligned = 1 when every group's row segment is a
// whole 128-byte line (Ts a multiple of 16 and a 128-byte aligned base): the kernel then needs no carries.
static inline int ap_prepare_stft16(ApStft16Params &W, const ApStftParams &P, int64_t B, int64_t Ts, int *grid,
                                    int *aligned) {
    if (P.L > (1 << 28)) return 1;                        // 32-bit sample offsets in the bounds-checked loads
    if (Ts < P.T) return 1;
    W.y = P.y;
    W.window = P.window;
    W.tw = P.tw;
    W.out = P.out_c;
    W.L = P.L;
    W.T = P.T;
    W.Ts = Ts;
    W.groups_per_clip = (P.T + APS16_G - 1) / APS16_G;
    W.n_groups = W.groups_per_clip * B;
    W.hop = P.hop;
    W.pad = P.pad;
    W.pad_mode = P.pad_mode;
    int off = APS_WAVES * APW_X_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += APW_TW2_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 16 * 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += APW_NC * (int)sizeof(ap_float2);
    *aligned = (Ts % APS16_G == 0 && (reinterpret_cast<uintptr_t>(P.out_c) & 127) == 0) ? 1 : 0;
    // the line-padded layout's whole-group tile (T2): its rows 576 .. 1024 here, the others over the exchange buffers;
    // otherwise two staging chunks
    static_assert(APS16_T2_SPLIT * APS16_OB_ROW <= APS_WAVES * APW_X_COMPLEX, "tile rows over the exchange buffers");
    const int ob_bytes = *aligned ? (1025 - APS16_T2_SPLIT) * APS16_OB_ROW * (int)sizeof(ap_float2)
                                  : 2 * APS16_OB_ROWS * APS16_OB_ROW * (int)sizeof(ap_float2);
    W.off_ob = off; off += ap_align16(ob_bytes);
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    if (Ts > (1 << 20)) return 1;                         // 32-bit row offsets (1024 Ts complex) in the store phase
    W.stagger = 0;
    W.gl_prev = nullptr; W.gl_mag = nullptr; W.gl_rebuilt = nullptr; W.gl_momentum = 0.0f;
    int64_t g = W.n_groups < 256 ? W.n_groups : 256;      // persistent: one workgroup per CU
    *grid = (int)g;
    return AP_OK;
}

// n_fft = 2048 fused ISTFT with 16-frame loads (kernels_istft16.h).  Ts = complex values between the rows of S.
// Returns 1 when the kernel does not apply (shape, LDS).
static inline int ap_prepare_istft16(ApIstft16Params &W, const float *S, const float *tw, int64_t B, int64_t T,
                                     int64_t Ts, const float *window, int hop, int64_t out_offset, int64_t out_len,
                                     float *y, int *grid) {
    if (B <= 0 || T <= 0 || Ts < T || Ts > 490000) return 1;          // a clip is one buffer resource: 1025 Ts 8 < 0xF0000000 bytes
    if (hop < 256 || hop > 1024 || 2048 % hop != 0) return 1;           // 256, 512, 1024 (hop = n_fft: the unfused kernels)
    if (out_offset % 4 != 0) return 1;
    W.S = reinterpret_cast<const ap_float2 *>(S);
    W.tw = reinterpret_cast<const ap_float2 *>(tw);
    W.window = window;
    W.y = y;
    W.T = T;
    W.Ts = Ts;
    W.g16_per_clip = (T + APS16_G - 1) / APS16_G;
    W.n_g16 = W.g16_per_clip * B;
    W.out_offset = out_offset;
    W.out_len = out_len;
    W.hop = hop;
    int off = APS_WAVES * APW_X_COMPLEX * (int)sizeof(ap_float2);
    // the staging area follows the exchange buffers: the two-round staging pass runs its first 768 rows across both
    static_assert(6 * 128 * APS16_OB_ROW <= APS_WAVES * APW_X_COMPLEX + 2 * APS16_OB_ROWS * APS16_OB_ROW, "round one of the tile");
    static_assert(257 <= 2 * APS16_OB_ROWS, "round two of the tile");
    W.off_ib = off; off += ap_align16(2 * APS16_OB_ROWS * APS16_OB_ROW * (int)sizeof(ap_float2));
    W.off_tw2 = off; off += APW_TW2_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 16 * 64 * (int)sizeof(ap_float2);
    W.off_win = off; off += 2048 * (int)sizeof(float);
    W.off_inv = off; off += hop * (int)sizeof(float);
    W.off_carry = off; off += 2 * (2048 - hop) * (int)sizeof(float);
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    // persistent, one workgroup per CU; a stretch that starts inside a clip re-runs 8 frames, so >= 4 steps (of 8
    // frames) each
    int64_t g = ((T + APS_WAVES - 1) / APS_WAVES) * B / 4;
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    *grid = (int)g;
    return AP_OK;
}

// One leg of the four-step transform.
static inline int ap_prepare_cfft_leg(ApCfftParams &C, int n, int64_t n_frames, int64_t B) {
    if (ap_make_cplan(n, &C.plan) != 0) AP_FAIL(AP_ERR_UNSUPPORTED, "cfft: cannot plan length %d", n);
    const int fstride = n + 1;
    const int64_t per_frame = (int64_t)2 * fstride * (int64_t)sizeof(ap_float2);
    if (per_frame > AP_LDS_MAX) AP_FAIL(AP_ERR_UNSUPPORTED, "cfft: length %d does not fit LDS", n);
    // 32 KB of LDS per workgroup = 4-5 workgroups per CU: the legs are latency-bound between their passes
    // (resample 256 x 220 500 -> 160 000: 3.75 ms at 64 KB, 2.50 at 32 KB, 2.86 at 16 KB; AP_CFFT_BUDGET overrides)
    static const int budget = std::getenv("AP_CFFT_BUDGET") ? std::atoi(std::getenv("AP_CFFT_BUDGET")) : 32 * 1024;
    int G = (int)(budget / per_frame);
    if (G < 1) G = 1;
    if (G > AP_MAX_G) G = AP_MAX_G;
    if ((int64_t)G > n_frames) G = (int)n_frames;
    C.tile.G = G;
    C.tile.fstride = fstride;
    C.tile.lds_bytes = (int)(per_frame * G);
    C.n_frames = n_frames;
    C.tiles_per_signal = (n_frames + G - 1) / G;
    if (C.tiles_per_signal * B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "cfft: grid too large");
    return AP_OK;
}

// Fill the two legs of a length-N complex transform (N = N1*N2) between `in` and `out` through
// the scratch array `mid` (all B x N complex unless real_in / real_out).
static inline int ap_prepare_cfft(ApCfftParams &L1, ApCfftParams &L2, const void *in, void *mid,
                                  void *out, int64_t B, int64_t N, int N1, int N2,
                                  const float *tw1, const float *tw2, int inverse, int real_in,
                                  int real_out, float scale) {
    int rc = ap_prepare_cfft_leg(L1, N1, N2, B);
    if (rc != AP_OK) return rc;
    rc = ap_prepare_cfft_leg(L2, N2, N1, B);
    if (rc != AP_OK) return rc;
    // leg 1: frames = n2 (N2 of them), index n1 at stride N2; out A[k1*N2 + n2], twiddle W_N^(n2*k1)
    L1.in = in; L1.out = mid;
    L1.in_batch = N; L1.out_batch = N;
    L1.in_fs = 1; L1.in_is = N2; L1.out_fs = 1; L1.out_is = N2;
    L1.tw_N = N2 > 1 ? N : 0;
    L1.tw = reinterpret_cast<const ap_float2 *>(tw1);
    L1.conj_io = inverse; L1.real_in = real_in; L1.real_out = 0; L1.scale = 1.0f;
    // leg 2: frames = k1 (N1 of them), contiguous rows of length N2; out X[k1 + N1*k2]
    L2.in = mid; L2.out = out;
    L2.in_batch = N; L2.out_batch = N;
    L2.in_fs = N2; L2.in_is = 1; L2.out_fs = 1; L2.out_is = N1;
    L2.tw_N = 0;
    L2.tw = reinterpret_cast<const ap_float2 *>(tw2);
    L2.conj_io = inverse; L2.real_in = 0; L2.real_out = real_out; L2.scale = scale;
    return AP_OK;
}

// One side (forward or inverse) of the FFT resampler: a length-N transform either directly on the
// four-step engine (M == 0: tw1 / tw2 are the leg tables of N) or as a chirp-z convolution of length M
// (tw1 / tw2: leg tables of M; chirp: N complex; spec: M complex).
struct ApCfftSide {
    int64_t N, M;
    const float *tw1, *tw2, *chirp, *spec;
};

// scipy.signal.resample for real rows = forward transform, spectrum surgery, inverse transform.  `Ops`
// launches the kernels (HIP in audioprims.hip, the CPU emulator in tests/emu): leg(ApCfftParams, B),
// spectrum(X, Nx, Y, num, B), chirp_pre(in, real_in, N, chirp, conj, out, M, B),
// chirp_spec(buf, spec, conj, M, B), chirp_post(buf, M, chirp, conj, N, scale, real_out, out, B).
// ws: two complex buffers of B * max(Nx, num, Mx, My) elements.
template <class Ops>
static inline int ap_cfft_inplace(Ops &ops, ap_float2 *buf, ap_float2 *mid, int64_t B, int64_t M, const float *tw1,
                                  const float *tw2, int inverse, float scale) {
    int m1, m2;
    if (ap_cfft_split(M, &m1, &m2) != 0) AP_FAIL(AP_ERR_UNSUPPORTED, "cfft: length %lld cannot be factored", (long long)M);
    ApCfftParams L1, L2;
    int rc = ap_prepare_cfft(L1, L2, buf, mid, buf, B, M, m1, m2, tw1, tw2, inverse, 0, 0, scale);
    if (rc != AP_OK) return rc;
    rc = ops.leg(L1, B);
    if (rc != AP_OK) return rc;
    return ops.leg(L2, B);
}

template <class Ops>
static inline int ap_resample_fft_compose(Ops &ops, const float *x, int64_t B, const ApCfftSide &X, const ApCfftSide &Y,
                                          float *ws, float *out) {
    const int64_t Nx = X.N, num = Y.N;
    int64_t Nmax = Nx > num ? Nx : num;
    if (X.M > Nmax) Nmax = X.M;
    if (Y.M > Nmax) Nmax = Y.M;
    ap_float2 *bufA = reinterpret_cast<ap_float2 *>(ws);
    ap_float2 *bufB = bufA + B * Nmax;
    ApCfftParams L1, L2;
    int rc;
    // ---- forward: x (real) -> bufB = X (B, Nx)
    if (X.M == 0) {
        int a1, a2;
        if (ap_cfft_split(Nx, &a1, &a2) != 0)
            AP_FAIL(AP_ERR_UNSUPPORTED, "resample(fft): length %lld has no factorisation N1*N2 with both <= %d",
                    (long long)Nx, AP_CFFT_MAX);
        rc = ap_prepare_cfft(L1, L2, x, bufA, bufB, B, Nx, a1, a2, X.tw1, X.tw2, 0, 1, 0, 1.0f);
        if (rc != AP_OK) return rc;
        if ((rc = ops.leg(L1, B)) != AP_OK) return rc;
        if ((rc = ops.leg(L2, B)) != AP_OK) return rc;
    } else {
        if (X.M < 2 * Nx - 1) AP_FAIL(AP_ERR_INVALID, "resample(fft): chirp length %lld < 2 N - 1", (long long)X.M);
        if ((rc = ops.chirp_pre(x, 1, Nx, X.chirp, 0, bufA, X.M, B)) != AP_OK) return rc;
        if ((rc = ap_cfft_inplace(ops, bufA, bufB, B, X.M, X.tw1, X.tw2, 0, 1.0f)) != AP_OK) return rc;
        if ((rc = ops.chirp_spec(bufA, X.spec, 0, X.M, B)) != AP_OK) return rc;
        if ((rc = ap_cfft_inplace(ops, bufA, bufB, B, X.M, X.tw1, X.tw2, 1, (float)(1.0 / (double)X.M))) != AP_OK) return rc;
        if ((rc = ops.chirp_post(bufA, X.M, X.chirp, 0, Nx, 1.0f, 0, bufB, B)) != AP_OK) return rc;
    }
    // ---- spectrum surgery: bufB (B, Nx) -> bufA (B, num), the full Hermitian spectrum
    if ((rc = ops.spectrum(bufB, Nx, bufA, num, B)) != AP_OK) return rc;
    // ---- inverse: bufA -> out (real), scale = (1 / num) * (num / Nx) = 1 / Nx
    const float scale = (float)(1.0 / (double)Nx);
    if (Y.M == 0) {
        int b1, b2;
        if (ap_cfft_split(num, &b1, &b2) != 0)
            AP_FAIL(AP_ERR_UNSUPPORTED, "resample(fft): length %lld has no factorisation N1*N2 with both <= %d",
                    (long long)num, AP_CFFT_MAX);
        rc = ap_prepare_cfft(L1, L2, bufA, bufB, out, B, num, b1, b2, Y.tw1, Y.tw2, 1, 0, 1, scale);
        if (rc != AP_OK) return rc;
        if ((rc = ops.leg(L1, B)) != AP_OK) return rc;
        return ops.leg(L2, B);
    }
    if (Y.M < 2 * num - 1) AP_FAIL(AP_ERR_INVALID, "resample(fft): chirp length %lld < 2 N - 1", (long long)Y.M);
    if ((rc = ops.chirp_pre(bufA, 0, num, Y.chirp, 1, bufB, Y.M, B)) != AP_OK) return rc;
    if ((rc = ap_cfft_inplace(ops, bufB, bufA, B, Y.M, Y.tw1, Y.tw2, 0, 1.0f)) != AP_OK) return rc;
    if ((rc = ops.chirp_spec(bufB, Y.spec, 1, Y.M, B)) != AP_OK) return rc;
    if ((rc = ap_cfft_inplace(ops, bufB, bufA, B, Y.M, Y.tw1, Y.tw2, 1, (float)(1.0 / (double)Y.M))) != AP_OK) return rc;
    return ops.chirp_post(bufB, Y.M, Y.chirp, 1, num, scale, 1, out, B);
}

static inline int ap_prepare_irfft_wave(ApIrfftWaveParams &W, const ApIrfftParams &P, int64_t B, int *grid) {
    W.S = P.S;
    W.tw = P.tw;
    W.frames = P.frames;
    W.T = P.T;
    W.groups_per_clip = (P.T + APS_WAVES - 1) / APS_WAVES;
    W.n_groups = W.groups_per_clip * B;
    int off = APS_WAVES * APW_X_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw2 = off; off += APW_TW2_COMPLEX * (int)sizeof(ap_float2);
    W.off_tw1 = off; off += 16 * 64 * (int)sizeof(ap_float2);
    W.off_ob = off; off += ap_align16(2 * APS_OB_ROWS * APS_OB_ROW * (int)sizeof(ap_float2));
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    if (W.T > (1 << 20)) return 1;                        // 32-bit row offsets (1024 T complex)
    int64_t g = W.n_groups < 256 ? W.n_groups : 256;
    *grid = (int)g;
    return AP_OK;
}

// fused irfft + overlap-add geometry (ap_istft_f32, n_fft = 2048): returns 1 when the fused kernel
// does not apply (hop shape, tiny problems where the one-group warm-up of a stretch would cost
// more than the separate overlap-add pass, LDS)
static inline bool ap_istft_fused_shape(int64_t B, int64_t T, int n_fft, int hop, int64_t out_offset) {
    if (n_fft != 2048 || B <= 0 || T <= 0 || T > (1 << 20)) return false;
    if (hop < 256 || hop > 1024 || 2048 % hop != 0) return false;      // 256, 512, 1024 (the gather shifts by log2 hop in {8, 9, 10})
    if (out_offset % 4 != 0) return false;
    return ((T + APS_WAVES - 1) / APS_WAVES) * B >= 64;
}

static inline int ap_prepare_istft_wave(ApIrfftWaveParams &W, const ApIrfftParams &P, int64_t B,
                                        const float *window, int hop, int64_t out_offset,
                                        int64_t out_len, float *y, int *grid) {
    int g0 = 0;
    if (!ap_istft_fused_shape(B, P.T, 2048, hop, out_offset)) return 1;
    if (ap_prepare_irfft_wave(W, P, B, &g0) != AP_OK) return 1;
    W.window = window;
    W.y = y;
    W.hop = hop;
    W.out_offset = out_offset;
    W.out_len = out_len;
    int off = W.lds_bytes;
    W.off_win = off; off += 2048 * (int)sizeof(float);
    W.off_carry = off; off += 2 * (2048 - hop) * (int)sizeof(float);
    W.lds_bytes = off;
    if (off > AP_LDS_MAX) return 1;
    int64_t g = W.n_groups / 4;                            // >= 4 groups per stretch
    if (g > 256) g = 256;
    *grid = (int)g;
    return AP_OK;
}

static inline int ap_prepare_irfft(ApIrfftParams &P, const float *S, int64_t B, int64_t T, int n_fft,
                                   const float *tw, float *frames) {
    if (!S || !tw || !frames) AP_FAIL(AP_ERR_INVALID, "irfft: NULL buffer");
    if (B <= 0 || T <= 0 || n_fft <= 0) AP_FAIL(AP_ERR_INVALID, "irfft: empty input");
    if (ap_make_plan(n_fft, &P.plan) != 0)
        AP_FAIL(AP_ERR_UNSUPPORTED, "n_fft=%d: cannot build FFT plan", n_fft);
    if (ap_make_tile(&P.plan, T, &P.tile) != 0)
        AP_FAIL(AP_ERR_UNSUPPORTED, "n_fft=%d does not fit the %d KiB LDS of one CU", n_fft,
                AP_LDS_MAX / 1024);
    P.S = reinterpret_cast<const ap_float2 *>(S);
    P.tw = reinterpret_cast<const ap_float2 *>(tw);
    P.frames = frames;
    P.T = T;
    P.tiles_per_clip = (T + P.tile.G - 1) / P.tile.G;
    P.n_bins = n_fft / 2 + 1;
    if (P.tiles_per_clip * B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "irfft: grid too large");
    return AP_OK;
}
