// libaudioprims_hip.so — C ABI entry points (include/audioprims.h): validate
// (ap_launch.h), then enqueue the gfx950 kernels.  Nothing here allocates device
// memory or synchronises.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>

#include <cstdio>
#include <cstring>

#include "ap_launch.h"
#include "ap_tu.h"
#include "kernels_generic.h"
#include "kernels_wave.h"
#include "kernels_pointwise.h"
#include "kernels_bigfft.h"
#include "kernels_ct.h"
#include "kernels_wave512.h"
#include "kernels_mel2048.h"
#include "kernels_frames8.h"
#include "kernels_features.h"

static thread_local char g_err[512] = "";

char *ap_error_buffer() { return g_err; }

void ap_set_error(const char *msg) { std::snprintf(g_err, sizeof(g_err), "%s", msg); }

static int ap_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return AP_OK;
}

template <class K>
static int ap_allow_lds(K kernel, int bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess)
            AP_FAIL(AP_ERR_HIP, "hipFuncSetAttribute(LDS=%d): %s", bytes, hipGetErrorString(e));
    }
    return AP_OK;
}

template <int PMODE, int PADGEN>
static int ap_launch_mel_wave(const ApMelWaveParams &W, int grid, void *stream) {
    int rc = ap_allow_lds(ap_mel2048_wave_kernel<PMODE, PADGEN>, W.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL((ap_mel2048_wave_kernel<PMODE, PADGEN>), dim3(grid), dim3(64 * APW_WAVES), W.lds_bytes,
                       (hipStream_t)stream, W);
    return ap_check_launch("ap_melspec_f32(wave)");
}

// n_fft = 2048 run kernel (kernels_mel2048.h)
template <int PMODE, int HOPJ, int IN16>
static int ap_launch_mel_run_h(const ApMelWaveParams &W, int n_pass, int grid, void *stream) {
#define AP_RUN_LAUNCH(NP)                                                                                \
    do {                                                                                                 \
        int rc = ap_allow_lds(ap_mel2048_run_kernel<PMODE, NP, HOPJ, IN16>, W.lds_bytes);                \
        if (rc != AP_OK) return rc;                                                                      \
        hipLaunchKernelGGL((ap_mel2048_run_kernel<PMODE, NP, HOPJ, IN16>), dim3(grid), dim3(64 * APM_WAVES), \
                           W.lds_bytes, (hipStream_t)stream, W);                                         \
    } while (0)
    if (n_pass == 1) AP_RUN_LAUNCH(1);
    else if (n_pass == 2) AP_RUN_LAUNCH(2);
    else if (n_pass == 3) AP_RUN_LAUNCH(3);
    else AP_RUN_LAUNCH(4);
#undef AP_RUN_LAUNCH
    return ap_check_launch("ap_melspec_f32(run)");
}
template <int PMODE, int IN16 = 0>
static int ap_launch_mel_run(const ApMelWaveParams &W, int n_pass, int grid, void *stream) {
    // hop = 512: the 1536 samples two consecutive frames share stay in registers
    return W.hopj == 4 ? ap_launch_mel_run_h<PMODE, 4, IN16>(W, n_pass, grid, stream)
                       : ap_launch_mel_run_h<PMODE, 0, IN16>(W, n_pass, grid, stream);
}

static const unsigned kApKeyMinusInf = 0x007FFFFFu;   // ap_fkey(-inf)

// compile-time specialised engine for n_fft = 400 / 512 / 1024 (kernels_ct.h)
template <int EPI, int PADGEN>
static int ap_launch_ct(ApStftParams &P, int n_fft, int64_t B, void *stream, bool *handled) {
    *handled = false;
    if (P.L > (1 << 28)) return AP_OK;                   // 32-bit sample offsets in the bounds-checked loads: generic engine
    int G = 0, lds = 0;
    *handled = false;
    if (!ap_ct_config(n_fft, EPI == 1 ? P.n_parts : 0, EPI == 1 ? P.n_quads : 0, P.n_mels, &G, &lds)) return AP_OK;
    P.tiles_per_clip = (P.T + G - 1) / G;
    int64_t tiles = P.tiles_per_clip * B;
    const int per_cu = AP_LDS_MAX / lds < 8 ? AP_LDS_MAX / lds : 8;
    int64_t grid = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    int rc = AP_OK;
#define AP_CT_LAUNCH(NC, R0, R1, R2, GG, NT)                                                             \
    do {                                                                                                 \
        rc = ap_allow_lds(ap_stft_ct_kernel<EPI, NC, R0, R1, R2, GG, PADGEN, NT>, lds);                  \
        if (rc != AP_OK) return rc;                                                                      \
        hipLaunchKernelGGL((ap_stft_ct_kernel<EPI, NC, R0, R1, R2, GG, PADGEN, NT>), dim3((unsigned)grid), \
                           dim3(NT), lds, (hipStream_t)stream, P);                                       \
    } while (0)
    if (n_fft == 400) AP_CT_LAUNCH(200, 8, 5, 5, 8, 256);
    else if (n_fft == 512) AP_CT_LAUNCH(256, 16, 16, 1, 8, 256);
    else AP_CT_LAUNCH(512, 16, 8, 4, 8, 256);
#undef AP_CT_LAUNCH
    *handled = true;
    return ap_check_launch("ap_stft_ct");
}

template <int PADGEN>
static int ap_launch_mel_wave_p(const ApMelWaveParams &W, int grid, float power, void *stream) {
    if (power == 2.0f) return ap_launch_mel_wave<2, PADGEN>(W, grid, stream);
    if (power == 1.0f) return ap_launch_mel_wave<1, PADGEN>(W, grid, stream);
    return ap_launch_mel_wave<0, PADGEN>(W, grid, stream);
}

template <int R>
static ApFrames8Geom ap_frames8_geom() {
    ApFrames8Geom G = {R, ApqGeom<R>::BS, ApqGeom<R>::PS, ApqGeom<R>::WMAX, ApqGeom<R>::WIN_REGS ? 0 : 1};
    return G;
}

// n_fft = 16 R mel-spectrogram on the eight-frames-per-wave kernel; *handled = false when it does not apply
template <int R>
static int ap_launch_mel8(const ApStftParams &P, int64_t B, const int32_t *plan, const int32_t *desc, float power,
                          uint32_t *max_key_dev, void *stream, bool *handled, int64_t Ts = 0) {
    ApFrames8Params W;
    int grid = 0;
    *handled = false;
    if (ap_prepare_frames8(W, P, B, true, plan, desc, APQ_WAVES, ap_frames8_geom<R>(), &grid) != AP_OK) return AP_OK;
    if (Ts > 0) {
        if (Ts < P.T) return AP_OK;
        W.Ts = Ts;
    }
    // Measured on one box: with the lane transpose (eight adjacent lanes per 32-byte run) Whisper 0.1436 ms, without
    // 0.1399; mel 512 0.2523 vs 0.2482 - the permutes cost more than the merged requests save.  Off unless asked for.
    static const bool transposed = std::getenv("AP_MEL8_TRANSPOSED_STORES") != nullptr;
    W.plain_stores = transposed ? 0 : 1;
    // An output tile in LDS so that a group's rows leave as 16-byte stores (a quarter of the store instructions):
    // same box, Whisper 0.1437 ms with it, 0.1392 without; mel 512 0.2488 vs 0.2436.  Off unless asked for.
    static const bool use_stage = std::getenv("AP_MEL8_STAGE") != nullptr;
    if (!use_stage) W.off_stage = 0;
    if (max_key_dev) {
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)max_key_dev, (int)kApKeyMinusInf, 1, (hipStream_t)stream);
        if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipMemsetD32Async: %s", hipGetErrorString(e));
        W.max_key = max_key_dev;
    }
    const bool padgen = !ap_clip_loads_ok(P);             // reflect / edge padding, odd hops
    auto kern = padgen ? (power == 2.0f ? ap_mel8_wave_kernel<R, 2, 1> : power == 1.0f ? ap_mel8_wave_kernel<R, 1, 1>
                                                                                        : ap_mel8_wave_kernel<R, 0, 1>)
                       : (power == 2.0f ? ap_mel8_wave_kernel<R, 2> : power == 1.0f ? ap_mel8_wave_kernel<R, 1>
                                                                                    : ap_mel8_wave_kernel<R, 0>);
    int rc = ap_allow_lds(kern, W.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APQ_WAVES), W.lds_bytes, (hipStream_t)stream, W);
    *handled = true;
    return ap_check_launch("ap_melspec_f32(frames8)");
}

template <int R>
static int ap_launch_stft8(const ApStftParams &P, int64_t B, void *stream, bool *handled, int64_t Ts = 0) {
    ApFrames8Params W;
    int grid = 0;
    *handled = false;
    if (ap_prepare_frames8(W, P, B, false, nullptr, nullptr, APQ_WAVES, ap_frames8_geom<R>(), &grid) != AP_OK) return AP_OK;
    if (Ts > 0) {                                          // padded rows: a clip's rows still have to fit 32-bit offsets
        if (Ts < P.T || Ts > (1 << 19)) return AP_OK;
        W.Ts = Ts;
    }
    auto kern = ap_clip_loads_ok(P) ? ap_stft8_wave_kernel<R, 0> : ap_stft8_wave_kernel<R, 1>;
    int rc = ap_allow_lds(kern, W.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APQ_WAVES), W.lds_bytes, (hipStream_t)stream, W);
    *handled = true;
    return ap_check_launch("ap_stft_f32(frames8)");
}

template <int PMODE, int HOPJ, int FLAT, int PGEN>
static int ap_launch_spec_run4(const ApSpecWaveParams &W, int grid, void *stream) {
    int rc = ap_allow_lds(ap_spec2048_run_kernel<PMODE, HOPJ, FLAT, PGEN>, W.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL((ap_spec2048_run_kernel<PMODE, HOPJ, FLAT, PGEN>), dim3(grid), dim3(64 * APM_WAVES), W.lds_bytes,
                       (hipStream_t)stream, W);
    return ap_check_launch("ap_spectral_audio_f32");
}

// the common shapes get their own instantiation; power not in {1, 2}, a bandwidth exponent != 2 and other
// hops share the general ones
template <int PMODE, int FLAT>
static int ap_launch_spec_run(const ApSpecWaveParams &W, int grid, void *stream) {
    const bool pgen = W.bandwidth && W.p != 2.0f;
    if (W.hopj == 4) return pgen ? ap_launch_spec_run4<PMODE, 4, FLAT, 1>(W, grid, stream)
                                 : ap_launch_spec_run4<PMODE, 4, FLAT, 0>(W, grid, stream);
    return pgen ? ap_launch_spec_run4<PMODE, 0, FLAT, 1>(W, grid, stream)
                : ap_launch_spec_run4<PMODE, 0, FLAT, 0>(W, grid, stream);
}

template <int R>
static int ap_launch_irfft8(const ApIrfftParams &P, int64_t B, void *stream, bool *handled) {
    ApIrfft8Params W;
    int grid = 0;
    *handled = false;
    if (ap_prepare_irfft8(W, P, B, R, ApqGeom<R>::BS, APQ_WAVES, &grid) != AP_OK) return AP_OK;
    int rc = ap_allow_lds(ap_irfft8_wave_kernel<R>, W.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_irfft8_wave_kernel<R>, dim3(grid), dim3(64 * APQ_WAVES), W.lds_bytes, (hipStream_t)stream, W);
    *handled = true;
    return ap_check_launch("ap_irfft_frames_f32(frames8)");
}

extern "C" {

int ap_version(void) { return 100; }

#ifdef AP_DIAG_STAMPS
// diagnostic build only: copy the per-wave (shader cycles, 100 MHz ticks) stamps to the host
int ap_diag_read_stamps(unsigned long long *host, int n_words) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(ap_diag_stamps), sizeof(unsigned long long) * (size_t)n_words);
    return e == hipSuccess ? AP_OK : AP_ERR_HIP;
}
#endif

const char *ap_last_error(void) { return g_err; }

int ap_pad_f32(const float *x, int64_t B, int64_t L, int64_t pad, int mode, float *out,
               void *stream) {
    int grid;
    int rc = ap_prepare_pad(x, B, L, pad, mode, out, &grid);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_pad_kernel, dim3(grid), dim3(AP_BLOCK), 0, (hipStream_t)stream, x, B, L,
                       pad, mode, out);
    return ap_check_launch("ap_pad_f32");
}

int ap_frame_f32(const float *x, int64_t B, int64_t L, int frame_length, int hop, float *out,
                 void *stream) {
    int grid;
    int64_t T;
    int rc = ap_prepare_frame(x, B, L, frame_length, hop, out, &T, &grid);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_frame_kernel, dim3(grid), dim3(AP_BLOCK), 0, (hipStream_t)stream, x, B, L,
                       T, frame_length, hop, out);
    return ap_check_launch("ap_frame_f32");
}

int ap_overlap_add_f32(const float *frames, const float *window, int64_t B, int64_t T, int n_fft,
                       int hop, int64_t out_offset, int64_t out_len, float *out, void *stream) {
    int64_t bpr;
    int rc = ap_prepare_ola(frames, window, B, T, n_fft, hop, out_offset, out_len, out, &bpr);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_overlap_add_kernel, dim3((unsigned)(bpr * B)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, frames, window, T, n_fft, hop, out_offset, out_len, bpr,
                       out);
    return ap_check_launch("ap_overlap_add_f32");
}

int ap_stft_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                const float *tw, int center, int pad_mode, int64_t T, float *out, void *stream) {
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, n_fft, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    if (!out) AP_FAIL(AP_ERR_INVALID, "stft: NULL output");
    P.out_c = reinterpret_cast<ap_float2 *>(out);
    if (n_fft == 2048) {
        // 16 frames per group: rows leave as 128-byte line-aligned windows (kernels_stft16.h);
        // AP_STFT2048_G8=1 keeps the 8-frame kernel of rounds 1-2 for A/B measurements
        static const bool g8 = std::getenv("AP_STFT2048_G8") != nullptr;
        if (!g8) {
            rc = ap_launch_stft16(P, B, T, stream);
            if (rc != 1) return rc;
        }
        ApStftWaveParams W;
        int grid = 0;
        if (ap_prepare_stft_wave(W, P, B, &grid) == AP_OK) {
            if (ap_clip_loads_ok(W)) {
                rc = ap_allow_lds(ap_stft2048_wave_kernel<0>, W.lds_bytes);
                if (rc != AP_OK) return rc;
                hipLaunchKernelGGL(ap_stft2048_wave_kernel<0>, dim3(grid), dim3(64 * APS_WAVES), W.lds_bytes,
                                   (hipStream_t)stream, W);
            } else {
                rc = ap_allow_lds(ap_stft2048_wave_kernel<1>, W.lds_bytes);
                if (rc != AP_OK) return rc;
                hipLaunchKernelGGL(ap_stft2048_wave_kernel<1>, dim3(grid), dim3(64 * APS_WAVES), W.lds_bytes,
                                   (hipStream_t)stream, W);
            }
            return ap_check_launch("ap_stft_f32(wave)");
        }
    }
    if (n_fft == 1024) {
        ApStftWave512Params W;
        int grid = 0;
        if (ap_prepare_stft_wave512(W, P, B, APHS_WAVES, APH_X_COMPLEX, APHS_OB_ROWS * APHS_OB_ROW, &grid) == AP_OK) {
            auto kern = W.padgen ? ap_stft1024_wave_kernel<1> : ap_stft1024_wave_kernel<0>;
            rc = ap_allow_lds(kern, W.lds_bytes);
            if (rc != AP_OK) return rc;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APHS_WAVES), W.lds_bytes, (hipStream_t)stream, W);
            return ap_check_launch("ap_stft_f32(wave512)");
        }
    }
    if (n_fft == 400 || n_fft == 512 || n_fft == 256) {     // eight frames per wave (kernels_frames8.h)
        static const bool force_ct = std::getenv("AP_STFT8_CT") != nullptr;      // A/B switch: keep the LDS engine
        if (!force_ct) {
            bool handled = false;
            rc = n_fft == 400 ? ap_launch_stft8<25>(P, B, stream, &handled)
                 : n_fft == 512 ? ap_launch_stft8<32>(P, B, stream, &handled)
                                : ap_launch_stft8<16>(P, B, stream, &handled);
            if (rc != AP_OK || handled) return rc;
        }
    }
    {
        bool handled = false;
        rc = ap_clip_loads_ok(P) ? ap_launch_ct<0, 0>(P, n_fft, B, stream, &handled)
                                                           : ap_launch_ct<0, 1>(P, n_fft, B, stream, &handled);
        if (rc != AP_OK || handled) return rc;
    }
    rc = ap_allow_lds(ap_stft_generic_kernel<0>, P.tile.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_stft_generic_kernel<0>, dim3((unsigned)(P.tiles_per_clip * B)),
                       dim3(AP_BLOCK), P.tile.lds_bytes, (hipStream_t)stream, P);
    return ap_check_launch("ap_stft_f32");
}

int ap_stft_rows_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                     const float *tw, int center, int pad_mode, int64_t T, int64_t row_stride, float *out, void *stream) {
    if (row_stride == T) return ap_stft_f32(y, B, L, n_fft, hop, window, tw, center, pad_mode, T, out, stream);
    if (row_stride < T) AP_FAIL(AP_ERR_INVALID, "stft: row_stride (%lld) must be >= the number of frames (%lld)",
                                (long long)row_stride, (long long)T);
    if (n_fft != 2048 && n_fft != 512 && n_fft != 400 && n_fft != 256)
        AP_FAIL(AP_ERR_UNSUPPORTED, "stft: padded rows are served for n_fft = 2048, 512, 400 and 256 only");
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, n_fft, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    if (!out) AP_FAIL(AP_ERR_INVALID, "stft: NULL output");
    P.out_c = reinterpret_cast<ap_float2 *>(out);
    if (n_fft != 2048) {                                   // eight frames per wave (kernels_frames8.h): 64-byte runs
        bool handled = false;
        rc = n_fft == 400 ? ap_launch_stft8<25>(P, B, stream, &handled, row_stride)
             : n_fft == 512 ? ap_launch_stft8<32>(P, B, stream, &handled, row_stride)
                            : ap_launch_stft8<16>(P, B, stream, &handled, row_stride);
        if (rc != AP_OK || handled) return rc;
        AP_FAIL(AP_ERR_UNSUPPORTED, "stft: shape not served with padded rows");
    }
    rc = ap_launch_stft16(P, B, row_stride, stream);
    if (rc == 1) AP_FAIL(AP_ERR_UNSUPPORTED, "stft: shape not served with padded rows");
    return rc;
}

int ap_melspec_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                   const float *tw, int center, int pad_mode, int64_t T, const float *fb,
                   const int32_t *plan, const int32_t *desc, int n_mels, float power, float *out,
                   void *stream) {
    return ap_melspec_max_f32(y, B, L, n_fft, hop, window, tw, center, pad_mode, T, fb, plan, desc, n_mels,
                              power, out, nullptr, stream);
}

int ap_melspec_max_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                       const float *tw, int center, int pad_mode, int64_t T, const float *fb,
                       const int32_t *plan, const int32_t *desc, int n_mels, float power, float *out,
                       uint32_t *max_key_dev, void *stream) {
    return ap_melspec_rows_f32(y, B, L, n_fft, hop, window, tw, center, pad_mode, T, T, fb, plan, desc, n_mels, power, out,
                               max_key_dev, stream);
}

int ap_melspec_rows_fused(int n_fft, int hop, int center, int pad_mode, int n_mels, float power, const int32_t *plan,
                          const int32_t *desc) {
    (void)hop; (void)center; (void)pad_mode;              // every padding mode has an instantiation of these kernels
    if (n_fft == 1024)                                    // kernels_wave512.h
        return (plan && desc && (desc[0] & AP_PLAN_PARTS) && !(desc[0] & AP_PLAN_FORCE_GENERIC) && n_mels <= 128) ? 1 : 0;
    if (n_fft == 400 || n_fft == 512 || n_fft == 256) {   // eight frames per wave (kernels_frames8.h)
        if (std::getenv("AP_MEL400_CT")) return 0;
        return (plan && desc && (desc[0] & AP_PLAN_BANDED) && (desc[0] & AP_PLAN_PARTS) && !(desc[0] & AP_PLAN_FORCE_GENERIC) &&
                n_mels <= 128 && desc[15] <= 4) ? 1 : 0;
    }
    static const bool force_wave = std::getenv("AP_MEL2048_WAVE") != nullptr;
    if (force_wave || n_fft != 2048 || !(power == 2.0f || power == 1.0f) || !ap_mel_wave_eligible(n_fft, plan, desc)) return 0;
    return (n_mels <= 128 && desc && desc[12] <= 256 && desc[15] <= 4) ? 1 : 0;
}

int ap_melspec_rows_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                        const float *tw, int center, int pad_mode, int64_t T, int64_t row_stride, const float *fb,
                        const int32_t *plan, const int32_t *desc, int n_mels, float power, float *out,
                        uint32_t *max_key_dev, void *stream) {
    if (row_stride < T) AP_FAIL(AP_ERR_INVALID, "melspectrogram: row_stride (%lld) must be >= the number of frames (%lld)",
                                (long long)row_stride, (long long)T);
    const bool rows = row_stride != T;
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, n_fft, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    rc = ap_prepare_mel(P, fb, plan, desc, n_mels, power, out);
    if (rc != AP_OK) return rc;
    if (ap_mel_wave_eligible(n_fft, plan, desc)) {
        ApMelWaveParams W;
        int grid = 0, n_pass = 0;
        // power 2 / 1 with constant padding and <= 128 filters: the run kernel (kernels_mel2048.h);
        // AP_MEL2048_WAVE=1 keeps the tile kernel for A/B timing (tools/mel_ab.py)
        static const bool force_wave = std::getenv("AP_MEL2048_WAVE") != nullptr;
        if (!force_wave && (power == 2.0f || power == 1.0f) &&
            ap_prepare_mel_run(W, P, B, plan, desc, APM_WAVES, APW_X_COMPLEX, APM_PARTIAL_OFF, &n_pass, &grid) == AP_OK) {
            W.Ts = row_stride;
            if (max_key_dev) {
                hipError_t e = hipMemsetD32Async((hipDeviceptr_t)max_key_dev, (int)kApKeyMinusInf, 1, (hipStream_t)stream);
                if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipMemsetD32Async: %s", hipGetErrorString(e));
                W.max_key = max_key_dev;
            }
            if (!ap_clip_loads_ok(P))      // reflect / edge padding, odd hops: index-remapped loads for the edge frames
                return power == 2.0f ? ap_launch_mel_run<2, 2>(W, n_pass, grid, stream)
                                     : ap_launch_mel_run<1, 2>(W, n_pass, grid, stream);
            return power == 2.0f ? ap_launch_mel_run<2>(W, n_pass, grid, stream)
                                 : ap_launch_mel_run<1>(W, n_pass, grid, stream);
        }
        if (rows) AP_FAIL(AP_ERR_UNSUPPORTED, "melspectrogram: padded rows are served by the n_fft = 2048 run kernel only");
        if (ap_prepare_mel_wave(W, P, B, plan, desc, &grid) == AP_OK) {
            if (max_key_dev) {            // the kernel raises the key itself: one atomic per wave
                hipError_t e = hipMemsetD32Async((hipDeviceptr_t)max_key_dev, (int)kApKeyMinusInf, 1, (hipStream_t)stream);
                if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipMemsetD32Async: %s", hipGetErrorString(e));
                W.max_key = max_key_dev;
            }
            // constant padding (or no centring) needs no index remap: bounds-checked buffer loads
            if (ap_clip_loads_ok(W))
                return ap_launch_mel_wave_p<0>(W, grid, power, stream);
            return ap_launch_mel_wave_p<1>(W, grid, power, stream);
        }
    }
    if (n_fft == 400 || n_fft == 512 || n_fft == 256) {     // eight frames per wave (kernels_frames8.h)
        static const bool force_ct = std::getenv("AP_MEL400_CT") != nullptr;     // A/B switch: keep the LDS engine
        if (!force_ct) {
            bool handled = false;
            const int64_t ts = rows ? row_stride : 0;
            rc = n_fft == 400 ? ap_launch_mel8<25>(P, B, plan, desc, power, max_key_dev, stream, &handled, ts)
                 : n_fft == 512 ? ap_launch_mel8<32>(P, B, plan, desc, power, max_key_dev, stream, &handled, ts)
                                : ap_launch_mel8<16>(P, B, plan, desc, power, max_key_dev, stream, &handled, ts);
            if (rc != AP_OK || handled) return rc;
        }
    }

    if (n_fft == 1024) {
        ApMelWave512Params W;
        int grid = 0;
        if (ap_prepare_mel_wave512(W, P, B, plan, desc, APH_WAVES, APH_X_COMPLEX, APH_PASSES, &grid) == AP_OK) {
            W.Ts = row_stride;
            if (max_key_dev) {
                hipError_t e = hipMemsetD32Async((hipDeviceptr_t)max_key_dev, (int)kApKeyMinusInf, 1, (hipStream_t)stream);
                if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipMemsetD32Async: %s", hipGetErrorString(e));
                W.max_key = max_key_dev;
            }
            // hop = 256: the next frame reuses three quarters of this frame's samples in registers;
            // reflect / edge padding and odd hops: index-remapped loads for the frames at the clip ends
            const bool padgen = !ap_clip_loads_ok(P);
#define AP_M1024(PM) (hop == 256 ? (padgen ? ap_mel1024_wave_kernel<PM, 2, 1> : ap_mel1024_wave_kernel<PM, 2, 0>) \
                                 : (padgen ? ap_mel1024_wave_kernel<PM, 0, 1> : ap_mel1024_wave_kernel<PM, 0, 0>))
            auto kern = power == 2.0f ? AP_M1024(2) : power == 1.0f ? AP_M1024(1) : AP_M1024(0);
#undef AP_M1024
            rc = ap_allow_lds(kern, W.lds_bytes);
            if (rc != AP_OK) return rc;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APH_WAVES), W.lds_bytes, (hipStream_t)stream, W);
            return ap_check_launch("ap_melspec_f32(wave512)");
        }
    }
    if (rows) AP_FAIL(AP_ERR_UNSUPPORTED, "melspectrogram: padded rows are served by the wave kernels only (n_fft 2048, 1024, 512, 400, 256)");
    if (!(desc && (desc[0] & AP_PLAN_FORCE_GENERIC))) {
        bool handled = false;
        rc = ap_clip_loads_ok(P) ? ap_launch_ct<1, 0>(P, n_fft, B, stream, &handled)
                                                           : ap_launch_ct<1, 1>(P, n_fft, B, stream, &handled);
        if (rc != AP_OK) return rc;
        if (handled) return max_key_dev ? ap_reduce_max_f32(out, B * (int64_t)n_mels * T, max_key_dev, stream) : AP_OK;
    }
    rc = ap_allow_lds(ap_stft_generic_kernel<1>, P.tile.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_stft_generic_kernel<1>, dim3((unsigned)(P.tiles_per_clip * B)),
                       dim3(AP_BLOCK), P.tile.lds_bytes, (hipStream_t)stream, P);
    rc = ap_check_launch("ap_melspec_f32");
    if (rc != AP_OK || !max_key_dev) return rc;
    return ap_reduce_max_f32(out, B * (int64_t)n_mels * T, max_key_dev, stream);
}

int ap_irfft_frames_f32(const float *S, int64_t B, int64_t T, int n_fft, const float *tw,
                        float *frames, void *stream) {
    ApIrfftParams P;
    int rc = ap_prepare_irfft(P, S, B, T, n_fft, tw, frames);
    if (rc != AP_OK) return rc;
    if (n_fft == 2048) {
        ApIrfftWaveParams W;
        int grid = 0;
        if (ap_prepare_irfft_wave(W, P, B, &grid) == AP_OK) {
            rc = ap_allow_lds(ap_irfft2048_wave_kernel<0>, W.lds_bytes);
            if (rc != AP_OK) return rc;
            hipLaunchKernelGGL(ap_irfft2048_wave_kernel<0>, dim3(grid), dim3(64 * APS_WAVES), W.lds_bytes,
                               (hipStream_t)stream, W);
            return ap_check_launch("ap_irfft_frames_f32(wave)");
        }
    }
    if ((n_fft == 512 || n_fft == 400 || n_fft == 256) && !std::getenv("AP_IRFFT8_GENERIC")) {
        bool handled = false;
        rc = n_fft == 512 ? ap_launch_irfft8<32>(P, B, stream, &handled)
             : n_fft == 400 ? ap_launch_irfft8<25>(P, B, stream, &handled) : ap_launch_irfft8<16>(P, B, stream, &handled);
        if (rc != AP_OK || handled) return rc;
    }
    rc = ap_allow_lds(ap_irfft_generic_kernel, P.tile.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_irfft_generic_kernel, dim3((unsigned)(P.tiles_per_clip * B)),
                       dim3(AP_BLOCK), P.tile.lds_bytes, (hipStream_t)stream, P);
    return ap_check_launch("ap_irfft_frames_f32");
}

int64_t ap_istft_workspace_floats(int64_t B, int64_t T, int n_fft, int hop, int64_t out_offset) {
    if (B <= 0 || T <= 0 || n_fft <= 0) return 0;
    if (ap_istft_fused_shape(B, T, n_fft, hop, out_offset)) return 0;
    if (ap_istft1024_fused_shape(B, T, n_fft, hop, out_offset)) return 0;
    if (out_offset >= 0 && ap_istft8_lds_bytes(T, n_fft, hop, APQ_WAVES) >= 0 && !std::getenv("AP_ISTFT8_UNFUSED")) return 0;
    return B * T * (int64_t)n_fft;
}

int ap_istft_f32(const float *S, int64_t B, int64_t T, int n_fft, int hop, const float *window,
                 const float *tw, float *frames_ws, int64_t out_offset, int64_t out_len, float *out,
                 void *stream) {
    if (n_fft == 2048 && S && tw && window && out && hop > 0 && out_len > 0 && ap_istft_fused_shape(B, T, 2048, hop, out_offset)) {
        // 16-frame loads (kernels_istft16.h); AP_ISTFT2048_G8=1 keeps the 8-frame kernel of rounds 1-2 for A/B runs
        static const bool g8 = std::getenv("AP_ISTFT2048_G8") != nullptr;
        if (!g8) {
            const int rc16 = ap_launch_istft16(S, tw, B, T, T, window, hop, out_offset, out_len, out, stream);
            if (rc16 != 1) return rc16;
        }
    }
    if (n_fft == 2048 && window && out && hop > 0 && out_len > 0) {
        // fused irfft + overlap-add: the (B, T, n_fft) frames never reach HBM
        ApIrfftParams P;
        int rc0 = ap_prepare_irfft(P, S, B, T, n_fft, tw, out /* frames pointer unused */);
        if (rc0 != AP_OK) return rc0;
        ApIrfftWaveParams W;
        int grid = 0;
        if (ap_prepare_istft_wave(W, P, B, window, hop, out_offset, out_len, out, &grid) == AP_OK) {
            rc0 = ap_allow_lds(ap_irfft2048_wave_kernel<1>, W.lds_bytes);
            if (rc0 != AP_OK) return rc0;
            hipLaunchKernelGGL(ap_irfft2048_wave_kernel<1>, dim3(grid), dim3(64 * APS_WAVES), W.lds_bytes,
                               (hipStream_t)stream, W);
            return ap_check_launch("ap_istft_f32(fused)");
        }
    }
    if (n_fft == 1024 && S && tw && window && out && hop > 0 && out_len > 0) {
        ApIstftWave512Params W;
        int grid = 0;
        if (ap_prepare_istft_wave512(W, S, tw, B, T, window, hop, out_offset, out_len, out, APHS_WAVES,
                                     APH_X_COMPLEX, APHS_OB_ROWS * APHS_OB_ROW, &grid) == AP_OK) {
            int rc0 = ap_allow_lds(ap_istft1024_wave_kernel, W.lds_bytes);
            if (rc0 != AP_OK) return rc0;
            hipLaunchKernelGGL(ap_istft1024_wave_kernel, dim3(grid), dim3(64 * APHS_WAVES), W.lds_bytes,
                               (hipStream_t)stream, W);
            return ap_check_launch("ap_istft_f32(fused 1024)");
        }
    }
    if (S && tw && window && out && out_len > 0 && B > 0 && !std::getenv("AP_ISTFT8_UNFUSED")) {
        ApIstft8Params W;
        int grid = 0;
        if (ap_prepare_istft8(W, S, tw, B, T, n_fft, window, hop, out_offset, out_len, out, APQ_WAVES, &grid) == AP_OK) {
            auto kern = n_fft == 512 ? ap_istft8_wave_kernel<32> : n_fft == 400 ? ap_istft8_wave_kernel<25>
                                                                                 : ap_istft8_wave_kernel<16>;
            int rc0 = ap_allow_lds(kern, W.lds_bytes);
            if (rc0 != AP_OK) return rc0;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APQ_WAVES), W.lds_bytes, (hipStream_t)stream, W);
            return ap_check_launch("ap_istft_f32(fused frames8)");
        }
    }
    if (!frames_ws) AP_FAIL(AP_ERR_INVALID, "istft: NULL workspace");
    int rc = ap_irfft_frames_f32(S, B, T, n_fft, tw, frames_ws, stream);
    if (rc != AP_OK) return rc;
    return ap_overlap_add_f32(frames_ws, window, B, T, n_fft, hop, out_offset, out_len, out, stream);
}

int ap_istft_rows_f32(const float *S, int64_t B, int64_t T, int64_t row_stride, int n_fft, int hop, const float *window,
                      const float *tw, int64_t out_offset, int64_t out_len, float *out, void *stream) {
    if (!S || !tw || !window || !out) AP_FAIL(AP_ERR_INVALID, "istft: NULL buffer");
    if (row_stride < T) AP_FAIL(AP_ERR_INVALID, "istft: row_stride (%lld) must be >= the number of frames (%lld)",
                                (long long)row_stride, (long long)T);
    if (hop <= 0 || out_len <= 0) AP_FAIL(AP_ERR_UNSUPPORTED, "istft: bad hop / length");
    if (n_fft == 512 || n_fft == 400 || n_fft == 256) {    // fused ISTFT of the frames8 family
        ApIstft8Params W;
        int grid = 0;
        if (B > 0 && row_stride <= (1 << 19) &&
            ap_prepare_istft8(W, S, tw, B, T, n_fft, window, hop, out_offset, out_len, out, APQ_WAVES, &grid) == AP_OK) {
            W.Ts = row_stride;
            auto kern = n_fft == 512 ? ap_istft8_wave_kernel<32> : n_fft == 400 ? ap_istft8_wave_kernel<25>
                                                                                 : ap_istft8_wave_kernel<16>;
            int rc0 = ap_allow_lds(kern, W.lds_bytes);
            if (rc0 != AP_OK) return rc0;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APQ_WAVES), W.lds_bytes, (hipStream_t)stream, W);
            return ap_check_launch("ap_istft_rows_f32(fused frames8)");
        }
        AP_FAIL(AP_ERR_UNSUPPORTED, "istft: shape not served with padded rows");
    }
    if (n_fft != 2048) AP_FAIL(AP_ERR_UNSUPPORTED, "istft: padded rows are served for n_fft = 2048, 512, 400 and 256 only");
    const int rc = ap_launch_istft16(S, tw, B, T, row_stride, window, hop, out_offset, out_len, out, stream);
    if (rc == 1) AP_FAIL(AP_ERR_UNSUPPORTED, "istft: shape not served with padded rows (hop must divide 2048, hop >= 256)");
    return rc;
}

// kept outputs [n_pre_remove, n_pre_remove + n_out) of upfirdn(taps, x, up, down) with zeros outside x
static int ap_launch_resample_poly(const float *x, int64_t B, int64_t L, int up, int down, const float *taps,
                                   int n_taps, int n_pre_remove, int64_t n_out, float *out, void *stream) {
    int rc;
    const int64_t bpr = (n_out + AP_BLOCK - 1) / AP_BLOCK;
    if (bpr * B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "resample_poly: grid too large");
    int lds = 0, Q = 0;
    static const bool no_decim2 = std::getenv("AP_DECIM_R4") != nullptr;             // A/B switch: four outputs per thread
    if (!no_decim2 && ap_resample_decim2_eligible(up, down, n_taps, &Q, &lds)) {
        const int64_t per_block = AP_BLOCK * 2 * Q;
        const int64_t bprq = (n_out + per_block - 1) / per_block;
        if (bprq * B <= kApMaxGrid) {
            rc = ap_allow_lds(ap_resample_decim2_kernel<4>, lds);
            if (rc != AP_OK) return rc;
            hipLaunchKernelGGL(ap_resample_decim2_kernel<4>, dim3((unsigned)(bprq * B)), dim3(AP_BLOCK), lds,
                               (hipStream_t)stream, x, L, down, taps, n_taps, n_pre_remove, n_out, bprq, out);
            return ap_check_launch("ap_resample_poly_f32(decim2)");
        }
    }
    if (ap_resample_decim_eligible(up, down, n_taps, &Q, &lds)) {
        const int64_t per_block = AP_BLOCK * 4 * Q;
        const int64_t bprq = (n_out + per_block - 1) / per_block;
        if (bprq * B <= kApMaxGrid) {
            const int steps = (n_taps + down * 3 + 3) & ~3;
            static const bool no_unroll = std::getenv("AP_DECIM_NO_UNROLL") != nullptr;    // A/B switch
            auto kern = Q == 4 ? ap_resample_decim_kernel<4> : Q == 2 ? ap_resample_decim_kernel<2> : ap_resample_decim_kernel<1>;
            if (Q == 2 && steps == 72 && !no_unroll) kern = ap_resample_decim_kernel<2, 72>;
            rc = ap_allow_lds(kern, lds);
            if (rc != AP_OK) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)(bprq * B)), dim3(AP_BLOCK), lds,
                               (hipStream_t)stream, x, L, down, taps, n_taps, n_pre_remove, n_out, bprq, out);
            return ap_check_launch("ap_resample_poly_f32(decim)");
        }
    }
    if (up > 1 && !std::getenv("AP_RESAMPLE_NAIVE")) {     // polyphase table + input span in LDS
        const int K = (n_taps + up - 1) / up, KS = K | 1;
        const int64_t per_block = (int64_t)AP_BLOCK * AP_RSPL_R;
        const int64_t span = (per_block * down) / up + K + 4;
        const int64_t lds = ((int64_t)up * KS + span) * (int64_t)sizeof(float);
        const int64_t bprl = (n_out + per_block - 1) / per_block;
        if (lds <= 64 * 1024 && bprl * B <= kApMaxGrid) {
            rc = ap_allow_lds(ap_resample_poly_lds_kernel, (int)lds);
            if (rc != AP_OK) return rc;
            hipLaunchKernelGGL(ap_resample_poly_lds_kernel, dim3((unsigned)(bprl * B)), dim3(AP_BLOCK), (size_t)lds,
                               (hipStream_t)stream, x, L, up, down, taps, n_taps, n_pre_remove, n_out, bprl, K, KS,
                               (int)span, out);
            return ap_check_launch("ap_resample_poly_f32(lds)");
        }
    }
    hipLaunchKernelGGL(ap_resample_poly_kernel, dim3((unsigned)(bpr * B)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, x, L, up, down, taps, n_taps, n_pre_remove, n_out, bpr, out);
    return ap_check_launch("ap_resample_poly_f32");
}

int ap_resample_poly_f32(const float *x, int64_t B, int64_t L, int up, int down, const float *taps,
                         int n_taps, int n_pre_remove, int64_t n_out, float *out, void *stream) {
    int64_t bpr;
    int rc = ap_prepare_resample_poly(x, B, L, up, down, taps, n_taps, n_pre_remove, n_out, out, &bpr);
    if (rc != AP_OK) return rc;
    return ap_launch_resample_poly(x, B, L, up, down, taps, n_taps, n_pre_remove, n_out, out, stream);
}

int ap_extend_f32(const float *x, int64_t B, int64_t L, int64_t n_ext, int mode, float *out, void *stream) {
    if (!x || !out) AP_FAIL(AP_ERR_INVALID, "extend: NULL buffer");
    if (B <= 0 || L <= 0 || n_ext < 0) AP_FAIL(AP_ERR_INVALID, "extend: signal must be non-empty");
    if (mode < AP_EXT_CONSTANT || mode > AP_EXT_LINE) AP_FAIL(AP_ERR_INVALID, "extend: unknown mode %d", mode);
    if (L < 2 && (mode == AP_EXT_SMOOTH || mode == AP_EXT_REFLECT || mode == AP_EXT_ANTIREFLECT || mode == AP_EXT_LINE))
        AP_FAIL(AP_ERR_INVALID, "extend: this mode needs at least two samples");
    hipLaunchKernelGGL(ap_extend_kernel, dim3(ap_grid_1d(B * (L + 2 * n_ext), AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, x, B, L, n_ext, mode, out);
    return ap_check_launch("ap_extend_f32");
}

int64_t ap_resample_poly_pad_samples(int up, int down, int n_taps) {
    if (up <= 0 || down <= 0 || n_taps <= 0) return -1;
    // every kept output reads input samples within n_taps / up of the signal; a multiple of `down` keeps the
    // output phase: the extended signal's output n + P up / down is the signal's output n
    const int64_t reach = (n_taps + up - 1) / up + 2;
    return (reach + down - 1) / down * down;
}

int ap_resample_poly_padded_f32(const float *x, int64_t B, int64_t L, int up, int down, const float *taps,
                                int n_taps, int n_pre_remove, int64_t n_out, int mode, float *ws, float *out,
                                void *stream) {
    int64_t bpr;
    int rc = ap_prepare_resample_poly(x, B, L, up, down, taps, n_taps, n_pre_remove, n_out, out, &bpr);
    if (rc != AP_OK) return rc;
    if (mode == AP_EXT_CONSTANT)
        return ap_launch_resample_poly(x, B, L, up, down, taps, n_taps, n_pre_remove, n_out, out, stream);
    if (!ws) AP_FAIL(AP_ERR_INVALID, "resample_poly: workspace missing");
    const int64_t P = ap_resample_poly_pad_samples(up, down, n_taps);
    rc = ap_extend_f32(x, B, L, P, mode, ws, stream);
    if (rc != AP_OK) return rc;
    const int64_t skip = n_pre_remove + P * up / down;
    if (skip > INT32_MAX) AP_FAIL(AP_ERR_UNSUPPORTED, "resample_poly: ratio too large");
    return ap_launch_resample_poly(ws, B, L + 2 * P, up, down, taps, n_taps, (int)skip, n_out, out, stream);
}

int ap_resample_linear_f32(const float *x, int64_t B, int64_t L, int64_t n_out, double scale,
                           float *out, void *stream) {
    if (!x || !out) AP_FAIL(AP_ERR_INVALID, "resample: NULL buffer");
    if (B <= 0 || L <= 0 || n_out <= 0) AP_FAIL(AP_ERR_INVALID, "resample: empty signal");
    hipLaunchKernelGGL(ap_resample_linear_kernel, dim3(ap_grid_1d(B * n_out, AP_BLOCK, kApStreamGrid)),
                       dim3(AP_BLOCK), 0, (hipStream_t)stream, x, B, L, n_out, scale, out);
    return ap_check_launch("ap_resample_linear_f32");
}

int ap_gl_project_f32(int mode, const float *S, const float *angles, const float *R, int64_t TR,
                      int64_t BF, int64_t T, float momentum, float *tprev, float *rebuilt,
                      void *stream) {
    if (!S || !rebuilt) AP_FAIL(AP_ERR_INVALID, "griffinlim: NULL buffer");
    if (mode == 0 && !angles) AP_FAIL(AP_ERR_INVALID, "griffinlim: angles missing");
    if (mode == 1 && (!R || TR < 0)) AP_FAIL(AP_ERR_INVALID, "griffinlim: R missing");
    if (mode == 1 && momentum > 0.0f && !tprev) AP_FAIL(AP_ERR_INVALID, "griffinlim: tprev missing");
    if (mode != 0 && mode != 1) AP_FAIL(AP_ERR_INVALID, "griffinlim: bad mode");
    if (BF <= 0 || T <= 0) return AP_OK;
    hipLaunchKernelGGL(ap_gl_project_kernel, dim3(ap_grid_1d(BF * T, AP_BLOCK, kApStreamGrid)),
                       dim3(AP_BLOCK), 0, (hipStream_t)stream, mode, S, angles,
                       reinterpret_cast<const ap_float2 *>(R), TR, BF, T, momentum,
                       reinterpret_cast<ap_float2 *>(tprev), reinterpret_cast<ap_float2 *>(rebuilt));
    return ap_check_launch("ap_gl_project_f32");
}

int ap_pcg64_uniform_f32(uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo,
                         double low, double high, int64_t n, float *out, void *stream) {
    if (n < 0 || (n > 0 && !out)) AP_FAIL(AP_ERR_INVALID, "pcg64_uniform: bad buffer");
    if (n == 0) return AP_OK;
    hipLaunchKernelGGL(ap_pcg64_uniform_kernel, dim3(ap_grid_1d(n, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, (unsigned long long)state_hi, (unsigned long long)state_lo,
                       (unsigned long long)inc_hi, (unsigned long long)inc_lo, low, high - low, n, out);
    return ap_check_launch("ap_pcg64_uniform_f32");
}

int ap_griffinlim_f32(const float *S, const float *angles, int64_t B, int64_t T, int n_fft, int hop,
                      const float *window, const float *tw, int center, int pad_mode,
                      int64_t out_offset, int64_t y_len, int64_t TR, int n_iter, float momentum,
                      float *rebuilt, float *tprev, float *R, float *frames_ws, float *y, void *stream) {
    if (!S || !angles || !rebuilt || !tprev || !R || !frames_ws || !y)
        AP_FAIL(AP_ERR_INVALID, "griffinlim: NULL buffer");
    if (n_iter <= 0) AP_FAIL(AP_ERR_INVALID, "n_iter must be positive, got %d", n_iter);
    if (momentum < 0.0f || momentum >= 1.0f) AP_FAIL(AP_ERR_INVALID, "momentum must be in [0, 1)");
    const int64_t F = n_fft / 2 + 1;
    int rc = ap_gl_project_f32(0, S, angles, nullptr, 0, B * F, T, 0.0f, tprev, rebuilt, stream);
    if (rc != AP_OK) return rc;
    if (TR == T) {
        // raw STFT ping-pong: `tprev` starts as the initial estimate and then alternates with `R`
        // as the previous / current raw spectrum; the projection never writes tprev back
        float *raw[2] = {R, tprev};
        for (int it = 0; it < n_iter; ++it) {
            float *cur = raw[it & 1], *prev = raw[(it + 1) & 1];
            rc = ap_istft_f32(rebuilt, B, T, n_fft, hop, window, tw, frames_ws, out_offset, y_len, y, stream);
            if (rc != AP_OK) return rc;
            rc = ap_stft_f32(y, B, y_len, n_fft, hop, window, tw, center, pad_mode, TR, cur, stream);
            if (rc != AP_OK) return rc;
            hipLaunchKernelGGL(ap_gl_project2_kernel, dim3(ap_grid_1d(B * F * T, AP_BLOCK, kApStreamGrid)),
                               dim3(AP_BLOCK), 0, (hipStream_t)stream, S, reinterpret_cast<const ap_float2 *>(cur),
                               reinterpret_cast<const ap_float2 *>(prev), B * F * T, momentum,
                               reinterpret_cast<ap_float2 *>(rebuilt));
            rc = ap_check_launch("ap_griffinlim_f32(project)");
            if (rc != AP_OK) return rc;
        }
        return ap_istft_f32(rebuilt, B, T, n_fft, hop, window, tw, frames_ws, out_offset, y_len, y, stream);
    }
    for (int it = 0; it < n_iter; ++it) {
        rc = ap_istft_f32(rebuilt, B, T, n_fft, hop, window, tw, frames_ws, out_offset, y_len, y, stream);
        if (rc != AP_OK) return rc;
        rc = ap_stft_f32(y, B, y_len, n_fft, hop, window, tw, center, pad_mode, TR, R, stream);
        if (rc != AP_OK) return rc;
        rc = ap_gl_project_f32(1, S, nullptr, R, TR, B * F, T, momentum, tprev, rebuilt, stream);
        if (rc != AP_OK) return rc;
    }
    return ap_istft_f32(rebuilt, B, T, n_fft, hop, window, tw, frames_ws, out_offset, y_len, y, stream);
}


// Griffin-Lim with line-padded workspaces (n_fft = 2048, TR == T): every spectrum of the loop lives with its rows
// `row_stride` complex values apart, so the STFT writes and the ISTFT reads whole 128-byte lines
// (kernels_stft16.h ALIGNED = 1, kernels_istft16.h) and the projection moves 16 bytes per lane.
//
// One chain (init projection, n_iter x {istft, stft, projection}, final istft) over clips [0, B) on ONE stream:
static int ap_gl_rows_chain(const float *S, const float *angles, int64_t B, int64_t T, int64_t row_stride, int hop,
                            const float *window, const float *tw, int center, int pad_mode, int64_t out_offset,
                            int64_t y_len, int n_iter, float momentum, float *rebuilt, float *tprev, float *R, float *y,
                            bool fused_projection, hipStream_t stream) {
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, y_len, 2048, hop, window, tw, center, pad_mode, T);   // also checks TR == T
    if (rc != AP_OK) return rc;
    const int64_t rows = B * 1025;
    const int64_t pairs = rows * (row_stride / 2);
    const int grid = ap_grid_1d(pairs, AP_BLOCK, kApStreamGrid);
    auto project = [&](const float *ang, const float *cur, const float *prev, float *tp) {
        if (pairs < (int64_t(1) << 31))
            hipLaunchKernelGGL(ap_gl_rows_kernel<int>, dim3(grid), dim3(AP_BLOCK), 0, stream, S, ang,
                               reinterpret_cast<const ap_float2 *>(cur), reinterpret_cast<const ap_float2 *>(prev), rows,
                               (int)T, (int)row_stride, momentum, reinterpret_cast<ap_float2 *>(tp),
                               reinterpret_cast<ap_float2 *>(rebuilt));
        else
            hipLaunchKernelGGL(ap_gl_rows_kernel<int64_t>, dim3(grid), dim3(AP_BLOCK), 0, stream, S, ang,
                               reinterpret_cast<const ap_float2 *>(cur), reinterpret_cast<const ap_float2 *>(prev), rows,
                               (int)T, (int)row_stride, momentum, reinterpret_cast<ap_float2 *>(tp),
                               reinterpret_cast<ap_float2 *>(rebuilt));
        return ap_check_launch("ap_griffinlim_rows_f32(project)");
    };
    rc = project(angles, nullptr, nullptr, tprev);                    // rebuilt = tprev = S exp(i angles)
    if (rc != AP_OK) return rc;
    // raw STFT ping-pong: `tprev` starts as the initial estimate and then alternates with `R` as the previous /
    // current raw spectrum (S unit(initial estimate) = the initial estimate: S >= 0)
    float *raw[2] = {R, tprev};
    for (int it = 0; it < n_iter; ++it) {
        float *cur = raw[it & 1], *prev = raw[(it + 1) & 1];
        rc = ap_launch_istft16(rebuilt, tw, B, T, row_stride, window, hop, out_offset, y_len, y, stream);
        if (rc == 1) AP_FAIL(AP_ERR_UNSUPPORTED, "griffinlim: istft shape not served with padded rows");
        if (rc != AP_OK) return rc;
        P.out_c = reinterpret_cast<ap_float2 *>(cur);
        rc = fused_projection ? ap_launch_stft16_gl(P, B, row_stride, prev, S, momentum, rebuilt, stream) : 1;
        if (rc == 1) {
            rc = ap_launch_stft16(P, B, row_stride, stream);
            if (rc == 1) AP_FAIL(AP_ERR_UNSUPPORTED, "griffinlim: stft shape not served with padded rows");
            if (rc != AP_OK) return rc;
            rc = project(nullptr, cur, prev, nullptr);
        }
        if (rc != AP_OK) return rc;
    }
    rc = ap_launch_istft16(rebuilt, tw, B, T, row_stride, window, hop, out_offset, y_len, y, stream);
    if (rc == 1) AP_FAIL(AP_ERR_UNSUPPORTED, "griffinlim: istft shape not served with padded rows");
    return rc;
}

// Helper streams of the Griffin-Lim loop, one set per device, created on first use and kept for the life of the
// process (the only state this library holds).  Clips are independent, so the batch runs as AP_GL_STREAMS
// (default 2) chains of half the clips each: the three kernels of an iteration are small at Griffin-Lim sizes
// (45-80 us each at 64 x 5 s), each alternates compute phases with bursts of memory traffic, and the streaming
// projection pass needs neither LDS nor many registers - side by side, one chain's kernels fill the other's
// bubbles.  The caller's stream forks into the helpers and joins them again with events: no host
// synchronisation, and everything stays ordered with respect to the caller's stream.
struct ApGlStreams {
    hipStream_t s[4];
    hipEvent_t fork, join[4];
    bool ok;
};
static ApGlStreams *ap_gl_streams() {
    static ApGlStreams per_device[64];
    static bool made[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    ApGlStreams &g = per_device[dev];
    if (!made[dev]) {
        made[dev] = true;
        g.ok = hipEventCreateWithFlags(&g.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 4 && g.ok; ++i)
            g.ok = hipStreamCreateWithFlags(&g.s[i], hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&g.join[i], hipEventDisableTiming) == hipSuccess;
    }
    return g.ok ? &g : nullptr;
}

int ap_griffinlim_rows_f32(const float *S, const float *angles, int64_t B, int64_t T, int64_t row_stride, int n_fft,
                           int hop, const float *window, const float *tw, int center, int pad_mode, int64_t out_offset,
                           int64_t y_len, int n_iter, float momentum, float *rebuilt, float *tprev, float *R, float *y,
                           void *stream) {
    if (!S || !angles || !rebuilt || !tprev || !R || !y || !window || !tw) AP_FAIL(AP_ERR_INVALID, "griffinlim: NULL buffer");
    if (n_iter <= 0) AP_FAIL(AP_ERR_INVALID, "n_iter must be positive, got %d", n_iter);
    if (momentum < 0.0f || momentum >= 1.0f) AP_FAIL(AP_ERR_INVALID, "momentum must be in [0, 1)");
    if (n_fft != 2048 || row_stride < T || (row_stride & 1) || row_stride > (1 << 20) ||
        ((reinterpret_cast<uintptr_t>(rebuilt) | reinterpret_cast<uintptr_t>(tprev) | reinterpret_cast<uintptr_t>(R)) & 15))
        AP_FAIL(AP_ERR_UNSUPPORTED, "griffinlim: padded rows need n_fft = 2048, an even row_stride >= T and 16-byte aligned workspaces");
    // A/B switches: AP_GL_STREAMS=1 keeps the whole batch on the caller's stream; AP_GL_PROJECT_PASS=1 runs the
    // projection as a pass of its own instead of in the STFT's store phase (kernels_stft16.h GL = 1).  Measured at
    // 64 x 5 s, 32 iterations: one stream 6.17 / 6.24 ms (separate / fused), two streams 5.64 / 5.44 ms.
    static const int n_streams_env = std::getenv("AP_GL_STREAMS") ? std::atoi(std::getenv("AP_GL_STREAMS")) : 2;
    static const bool fused = std::getenv("AP_GL_PROJECT_PASS") == nullptr;
    int n_sub = n_streams_env < 1 ? 1 : (n_streams_env > 4 ? 4 : n_streams_env);
    while (n_sub > 1 && !ap_istft_fused_shape(B / n_sub, T, 2048, hop, out_offset)) --n_sub;
    if (!ap_istft_fused_shape(B / n_sub, T, 2048, hop, out_offset)) AP_FAIL(AP_ERR_UNSUPPORTED, "griffinlim: shape not served with padded rows");
    ApGlStreams *gs = n_sub > 1 ? ap_gl_streams() : nullptr;
    if (!gs) n_sub = 1;
    if (n_sub == 1)
        return ap_gl_rows_chain(S, angles, B, T, row_stride, hop, window, tw, center, pad_mode, out_offset, y_len, n_iter,
                                momentum, rebuilt, tprev, R, y, fused, (hipStream_t)stream);
    if (hipEventRecord(gs->fork, (hipStream_t)stream) != hipSuccess) AP_FAIL(AP_ERR_HIP, "griffinlim: hipEventRecord failed");
    int rc = AP_OK;
    static const int cap_env = std::getenv("AP_GL_GRID_CAP") ? std::atoi(std::getenv("AP_GL_GRID_CAP")) : 0;
    struct CapGuard { int old; CapGuard(int v) : old(ap_g16_grid_cap) { ap_g16_grid_cap = v; } ~CapGuard() { ap_g16_grid_cap = old; } } guard(cap_env);
    const int64_t spec = 1025 * T, wsp = 1025 * row_stride * 2;
    for (int i = 0; i < n_sub; ++i) {
        const int64_t b0 = B * i / n_sub, b1 = B * (i + 1) / n_sub;
        if (hipStreamWaitEvent(gs->s[i], gs->fork, 0) != hipSuccess) AP_FAIL(AP_ERR_HIP, "griffinlim: hipStreamWaitEvent failed");
        const int rci = ap_gl_rows_chain(S + b0 * spec, angles + b0 * spec, b1 - b0, T, row_stride, hop, window, tw, center,
                                         pad_mode, out_offset, y_len, n_iter, momentum, rebuilt + b0 * wsp, tprev + b0 * wsp,
                                         R + b0 * wsp, y + b0 * y_len, fused, gs->s[i]);
        if (rci != AP_OK && rc == AP_OK) rc = rci;
        // join even after an error: the caller's stream must not run ahead of what was enqueued
        if (hipEventRecord(gs->join[i], gs->s[i]) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, gs->join[i], 0) != hipSuccess)
            AP_FAIL(AP_ERR_HIP, "griffinlim: joining the helper streams failed");
    }
    return rc;
}

// out[0] = mean((a - b)^2) over n floats; ws = ap_mse_workspace_doubles() float64 values of scratch
int64_t ap_mse_workspace_doubles(void) { return kApStreamGrid; }
int ap_mse_f32(const float *a, const float *b, int64_t n, double *ws, float *out, void *stream) {
    if (n <= 0 || !a || !b || !ws || !out) AP_FAIL(AP_ERR_INVALID, "mse: bad arguments");
    const int grid = ap_grid_1d(n, AP_BLOCK, kApStreamGrid);
    hipLaunchKernelGGL(ap_sqdiff_partial_kernel, dim3(grid), dim3(AP_BLOCK), 0, (hipStream_t)stream, a, b, n, ws);
    hipLaunchKernelGGL(ap_sum_partials_kernel, dim3(1), dim3(AP_BLOCK), 0, (hipStream_t)stream, ws, grid, 1.0 / (double)n, out);
    return ap_check_launch("ap_mse_f32");
}

int ap_reduce_max_f32(const float *x, int64_t n, uint32_t *key_dev, void *stream) {
    if (!x || !key_dev || n <= 0) AP_FAIL(AP_ERR_INVALID, "reduce_max: bad arguments");
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)key_dev, (int)kApKeyMinusInf, 1, (hipStream_t)stream);
    if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipMemsetD32Async: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(ap_reduce_max_kernel, dim3(ap_grid_1d(n, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK),
                       AP_BLOCK * sizeof(float), (hipStream_t)stream, x, n, key_dev);
    return ap_check_launch("ap_reduce_max_f32");
}

int ap_to_db_f32(const float *S, int64_t n, float coef, float amin, float ref_value,
                 const uint32_t *ref_key_dev, float top_db, float *out, uint32_t *ws_dev,
                 void *stream) {
    if (n < 0 || (n > 0 && (!S || !out))) AP_FAIL(AP_ERR_INVALID, "to_db: bad buffer");
    if (n == 0) return AP_OK;
    const bool clip = top_db >= 0.0f;
    if (clip && !ws_dev) AP_FAIL(AP_ERR_INVALID, "to_db: top_db needs a scratch word");
    if (clip) {
        int rc = ap_reduce_max_f32(S, n, ws_dev, stream);      // max(out) = dB(max(S)): see kernels_pointwise.h
        if (rc != AP_OK) return rc;
    }
    ApDbParams D;
    D.coef = coef; D.amin = amin; D.ref_value = ref_value; D.top_db = clip ? top_db : -1.0f;
    D.ref_key = ref_key_dev; D.smax_key = ws_dev;
    const int grid = ap_grid_1d(n, AP_BLOCK, kApStreamGrid);
    hipLaunchKernelGGL(ap_to_db_kernel, dim3(grid), dim3(AP_BLOCK), 0, (hipStream_t)stream, S, n, D, out);
    return ap_check_launch("ap_to_db_f32");
}

int ap_from_db_f32(const float *x, int64_t n, float ref, float div, float *out, void *stream) {
    if (n < 0 || (n > 0 && (!x || !out))) AP_FAIL(AP_ERR_INVALID, "from_db: bad buffer");
    if (n == 0) return AP_OK;
    hipLaunchKernelGGL(ap_from_db_kernel, dim3(ap_grid_1d(n, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, x, n, ref, div, out);
    return ap_check_launch("ap_from_db_f32");
}

}  // extern "C"

template <int KT, int DB, int WIDE>
static int ap_launch_dct_as(const float *x, const float *C, const float *row_scale, int64_t outer, int n_in,
                            int64_t inner, int n_out, const ApDbParams &D, float *out, void *stream, int lds) {
    const int grid = ap_grid_1d(outer * inner, AP_BLOCK, kApStreamGrid);
    // 16 outputs: held to five waves per SIMD (<= 96 VGPRs; HIP's second launch-bound is waves per SIMD); the pass
    // waits on its row loads, so the fifth wave is worth more than the handful of spilled words (mfcc tail of
    // config 4: 59.9 -> 55.4 us)
    constexpr int MINB = KT == 16 ? 5 : 1;
    int rc = ap_allow_lds(ap_dct_kernel<KT, DB, WIDE, MINB>, lds);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL((ap_dct_kernel<KT, DB, WIDE, MINB>), dim3(grid), dim3(AP_BLOCK), lds, (hipStream_t)stream, x, C,
                       row_scale, outer, n_in, inner, n_out, D, out);
    return ap_check_launch("ap_dct_f32");
}

template <int DB>
static int ap_launch_dct(const float *x, const float *C, const float *row_scale, int64_t outer, int n_in,
                         int64_t inner, int n_out, const ApDbParams &D, float *out, void *stream,
                         bool *handled) {
    *handled = false;
    const int KT = n_out <= 16 ? 16 : 32;
    const int lds = n_in * KT * (int)sizeof(float);
    if (lds > 64 * 1024) return AP_OK;
    *handled = true;
    const int64_t most = outer * inner * (n_in > n_out ? n_in : n_out);
    const bool wide = most >= ((int64_t)1 << 31);
    if (KT == 16)
        return wide ? ap_launch_dct_as<16, DB, 1>(x, C, row_scale, outer, n_in, inner, n_out, D, out, stream, lds)
                    : ap_launch_dct_as<16, DB, 0>(x, C, row_scale, outer, n_in, inner, n_out, D, out, stream, lds);
    return wide ? ap_launch_dct_as<32, DB, 1>(x, C, row_scale, outer, n_in, inner, n_out, D, out, stream, lds)
                : ap_launch_dct_as<32, DB, 0>(x, C, row_scale, outer, n_in, inner, n_out, D, out, stream, lds);
}

extern "C" {
int ap_dct_f32(const float *x, const float *C, const float *row_scale, int64_t outer, int n_in,
               int64_t inner, int n_out, float *out, void *stream) {
    if (!x || !C || !out) AP_FAIL(AP_ERR_INVALID, "dct: NULL buffer");
    if (n_in <= 0 || n_out <= 0) AP_FAIL(AP_ERR_INVALID, "dct: sizes must be positive");
    if (outer <= 0 || inner <= 0) return AP_OK;
    ApDbParams D = {};
    bool handled = false;
    int rc = ap_launch_dct<0>(x, C, row_scale, outer, n_in, inner, n_out, D, out, stream, &handled);
    if (rc != AP_OK || handled) return rc;
    const int grid = ap_grid_1d(outer * inner, AP_BLOCK, kApStreamGrid);
    if (n_out <= 16)
        hipLaunchKernelGGL(ap_dct_generic_kernel<16>, dim3(grid), dim3(AP_BLOCK), 0, (hipStream_t)stream, x, C,
                           row_scale, outer, n_in, inner, n_out, out);
    else
        hipLaunchKernelGGL(ap_dct_generic_kernel<32>, dim3(grid), dim3(AP_BLOCK), 0, (hipStream_t)stream, x, C,
                           row_scale, outer, n_in, inner, n_out, out);
    return ap_check_launch("ap_dct_f32");
}

int ap_db_dct_f32(const float *S, const float *C, const float *row_scale, int64_t outer, int n_in,
                  int64_t inner, int n_out, float coef, float amin, float ref_value,
                  const uint32_t *ref_key_dev, float top_db, uint32_t *ws_dev, int max_ready, float *out,
                  void *stream) {
    if (!S || !C || !out) AP_FAIL(AP_ERR_INVALID, "db_dct: NULL buffer");
    if (n_in <= 0 || n_out <= 0) AP_FAIL(AP_ERR_INVALID, "db_dct: sizes must be positive");
    if (outer <= 0 || inner <= 0) return AP_OK;
    const bool clip = top_db >= 0.0f;
    if (clip && !ws_dev) AP_FAIL(AP_ERR_INVALID, "db_dct: top_db needs a scratch word");
    if (n_in * (n_out <= 16 ? 16 : 32) * (int)sizeof(float) > 64 * 1024)
        AP_FAIL(AP_ERR_UNSUPPORTED, "db_dct: n_in=%d too long for the fused kernel (use ap_to_db_f32 + ap_dct_f32)", n_in);
    if (clip && !max_ready) {
        int rc = ap_reduce_max_f32(S, outer * n_in * inner, ws_dev, stream);
        if (rc != AP_OK) return rc;
    }
    ApDbParams D;
    D.coef = coef; D.amin = amin; D.ref_value = ref_value; D.top_db = clip ? top_db : -1.0f;
    D.ref_key = ref_key_dev; D.smax_key = ws_dev;
    bool handled = false;
    return ap_launch_dct<1>(S, C, row_scale, outer, n_in, inner, n_out, D, out, stream, &handled);
}

static int ap_launch_cfft_leg(const ApCfftParams &C, int64_t B, void *stream) {
    int rc = ap_allow_lds(ap_cfft_strided_kernel, C.tile.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_cfft_strided_kernel, dim3((unsigned)(C.tiles_per_signal * B)), dim3(AP_BLOCK),
                       C.tile.lds_bytes, (hipStream_t)stream, C);
    return ap_check_launch("ap_cfft leg");
}

int ap_cfft_split_host(int64_t N, int *N1, int *N2) {
    if (!N1 || !N2) return AP_ERR_INVALID;
    return ap_cfft_split(N, N1, N2) == 0 ? AP_OK : AP_ERR_UNSUPPORTED;
}

namespace {
struct ApHipFftOps {                      // kernel launches of ap_resample_fft_compose on a HIP stream
    void *stream;
    int leg(const ApCfftParams &C, int64_t B) { return ap_launch_cfft_leg(C, B, stream); }
    int spectrum(const ap_float2 *X, int64_t Nx, ap_float2 *Y, int64_t num, int64_t B) {
        hipLaunchKernelGGL(ap_resample_spectrum_kernel, dim3(ap_grid_1d(B * num, AP_BLOCK, kApStreamGrid)),
                           dim3(AP_BLOCK), 0, (hipStream_t)stream, X, Nx, Y, num, B);
        return ap_check_launch("ap_resample_spectrum");
    }
    int chirp_pre(const void *in, int real_in, int64_t N, const float *chirp, int conj, ap_float2 *out, int64_t M,
                  int64_t B) {
        hipLaunchKernelGGL(ap_chirp_pre_kernel, dim3(ap_grid_1d(B * M, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                           (hipStream_t)stream, in, real_in, N, reinterpret_cast<const ap_float2 *>(chirp), conj, out, M, B);
        return ap_check_launch("ap_chirp_pre");
    }
    int chirp_spec(ap_float2 *buf, const float *spec, int conj, int64_t M, int64_t B) {
        hipLaunchKernelGGL(ap_chirp_spec_kernel, dim3(ap_grid_1d(B * M, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                           (hipStream_t)stream, buf, reinterpret_cast<const ap_float2 *>(spec), conj, M, B);
        return ap_check_launch("ap_chirp_spec");
    }
    int chirp_post(const ap_float2 *buf, int64_t M, const float *chirp, int conj, int64_t N, float scale, int real_out,
                   void *out, int64_t B) {
        hipLaunchKernelGGL(ap_chirp_post_kernel, dim3(ap_grid_1d(B * N, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                           (hipStream_t)stream, buf, M, reinterpret_cast<const ap_float2 *>(chirp), conj, N, scale,
                           real_out, out, B);
        return ap_check_launch("ap_chirp_post");
    }
};
}  // namespace

int ap_resample_fft_f32(const float *x, int64_t B, int64_t Nx, int64_t num, const float *tw_x1,
                        const float *tw_x2, const float *tw_y1, const float *tw_y2, float *ws,
                        float *out, void *stream) {
    return ap_resample_fft_chirp_f32(x, B, Nx, num, 0, tw_x1, tw_x2, nullptr, nullptr, 0, tw_y1, tw_y2, nullptr,
                                     nullptr, ws, out, stream);
}

int ap_resample_fft_chirp_f32(const float *x, int64_t B, int64_t Nx, int64_t num, int64_t Mx, const float *tw_x1,
                              const float *tw_x2, const float *chirp_x, const float *spec_x, int64_t My,
                              const float *tw_y1, const float *tw_y2, const float *chirp_y, const float *spec_y,
                              float *ws, float *out, void *stream) {
    if (!x || !out || !ws || !tw_x1 || !tw_x2 || !tw_y1 || !tw_y2)
        AP_FAIL(AP_ERR_INVALID, "resample(fft): NULL buffer");
    if (B <= 0 || Nx <= 0 || num <= 0) AP_FAIL(AP_ERR_INVALID, "resample(fft): empty signal");
    if ((Mx > 0 && (!chirp_x || !spec_x)) || (My > 0 && (!chirp_y || !spec_y)) || Mx < 0 || My < 0)
        AP_FAIL(AP_ERR_INVALID, "resample(fft): chirp tables missing");
    const ApCfftSide X = {Nx, Mx, tw_x1, tw_x2, chirp_x, spec_x};
    const ApCfftSide Y = {num, My, tw_y1, tw_y2, chirp_y, spec_y};
    ApHipFftOps ops = {stream};
    return ap_resample_fft_compose(ops, x, B, X, Y, ws, out);
}

static int ap_complex_unary(const float *S, int64_t n, int mode, float *out, void *stream) {
    if (n < 0 || (n > 0 && (!S || !out))) AP_FAIL(AP_ERR_INVALID, "complex op: bad buffer");
    if (n == 0) return AP_OK;
    hipLaunchKernelGGL(ap_complex_unary_kernel, dim3(ap_grid_1d(n, AP_BLOCK, kApStreamGrid)),
                       dim3(AP_BLOCK), 0, (hipStream_t)stream,
                       reinterpret_cast<const ap_float2 *>(S), n, mode, out);
    return ap_check_launch("ap_complex_unary");
}

int ap_magnitude_f32(const float *S, int64_t n, float *out, void *stream) {
    return ap_complex_unary(S, n, 0, out, stream);
}

int ap_phase_f32(const float *S, int64_t n, float *out, void *stream) {
    return ap_complex_unary(S, n, 1, out, stream);
}

// |S| (mode 0) or atan2 (mode 1) of a spectrum with padded rows -> dense (rows, T) floats
int ap_complex_unary_rows_f32(const float *S, int64_t rows, int64_t T, int64_t row_stride, int mode, float *out,
                              void *stream) {
    if (rows < 0 || T < 0 || row_stride < T || (mode != 0 && mode != 1) || T > 2147483647LL || row_stride > 2147483647LL)
        AP_FAIL(AP_ERR_INVALID, "magnitude / phase: bad shape");
    if (rows == 0 || T == 0) return AP_OK;
    if (!S || !out) AP_FAIL(AP_ERR_INVALID, "magnitude / phase: NULL buffer");
    hipLaunchKernelGGL(ap_complex_unary_rows_kernel, dim3(ap_grid_1d(rows * T, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, reinterpret_cast<const ap_float2 *>(S), rows, (int)T, (int)row_stride, mode, out);
    return ap_check_launch("ap_complex_unary_rows_f32");
}

// ---------------------------------------------------------------------- §8(f): features / framing
int ap_spectral_stats_f32(const float *S, int is_complex, int64_t B, int64_t F, int64_t T, const float *freq,
                          float power, const float *centroid_in, float p, int norm, float roll_percent, float amin,
                          float *centroid, float *bandwidth, float *rolloff, float *flatness, void *stream) {
    if (!S || !freq) AP_FAIL(AP_ERR_INVALID, "spectral features: NULL buffer");
    if (B <= 0 || F <= 0 || T <= 0)
        AP_FAIL(AP_ERR_INVALID, "S must be 2D (freq_bins, n_frames) or 3D (batch, freq_bins, n_frames)");
    if (roll_percent < 0.0f || roll_percent > 1.0f) AP_FAIL(AP_ERR_INVALID, "roll_percent must be between 0 and 1");
    if (!(p > 0.0f)) AP_FAIL(AP_ERR_INVALID, "p must be positive");
    if (!centroid && !bandwidth && !rolloff && !flatness) return AP_OK;
    ApSpectralParams P;
    P.S = S; P.freq = freq; P.centroid_in = centroid_in;
    P.centroid = centroid; P.bandwidth = bandwidth; P.rolloff = rolloff; P.flatness = flatness;
    P.F = F; P.T = T; P.tiles_per_clip = (T + APF_TX - 1) / APF_TX;
    P.is_complex = is_complex; P.norm = norm;
    P.power = power; P.p = p; P.roll_percent = roll_percent; P.amin = amin;
    if (P.tiles_per_clip * B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "spectral features: grid too large");
    hipLaunchKernelGGL(ap_spectral_stats_kernel, dim3((unsigned)(P.tiles_per_clip * B)), dim3(APF_TX * APF_TY), 0,
                       (hipStream_t)stream, P);
    return ap_check_launch("ap_spectral_stats_f32");
}

int ap_spectral_audio_fused(int64_t L, int n_fft, int hop, int center, int pad_mode) {
    if (n_fft != 2048 || hop <= 0 || L <= 0) return 0;
    const int pad = center ? n_fft / 2 : 0;
    if (pad != 0 && (pad_mode != AP_PAD_CONSTANT || (hop & 1))) return 0;
    if (!center && L < n_fft) return 0;
    // the bounds ap_prepare_spec_run enforces (32-bit sample offsets, 24-bit frame counts): past them the caller
    // must take the two-kernel route instead of failing in the launch
    if (L > (1 << 28) || ap_n_frames(L, n_fft, hop, center) > (1 << 24)) return 0;
    return std::getenv("AP_SPEC_TWO_KERNELS") ? 0 : 1;      // A/B switch: keep the STFT + statistics route
}

int ap_spectral_audio_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                          const float *tw, int center, int pad_mode, int64_t T, const float *freq, float power,
                          float p, int norm, float roll_percent, float amin, float *centroid, float *bandwidth,
                          float *rolloff, float *flatness, void *stream) {
    if (!freq) AP_FAIL(AP_ERR_INVALID, "spectral features: NULL buffer");
    if (roll_percent < 0.0f || roll_percent > 1.0f) AP_FAIL(AP_ERR_INVALID, "roll_percent must be between 0 and 1");
    if (!(p > 0.0f)) AP_FAIL(AP_ERR_INVALID, "p must be positive");
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, n_fft, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    if (!centroid && !bandwidth && !rolloff && !flatness) return AP_OK;
    ApSpecWaveParams W;
    int grid = 0;
    if (!ap_spectral_audio_fused(L, n_fft, hop, center, pad_mode) ||
        ap_prepare_spec_run(W, P, B, APM_WAVES, APW_X_COMPLEX, &grid) != AP_OK)
        AP_FAIL(AP_ERR_UNSUPPORTED, "spectral features from audio: shape not served by the fused kernel");
    W.freq = freq;
    W.centroid = centroid; W.bandwidth = bandwidth; W.rolloff = rolloff; W.flatness = flatness;
    W.power = power; W.p = p; W.norm = norm; W.roll_percent = roll_percent; W.amin = amin;
    if (flatness)
        return power == 2.0f ? ap_launch_spec_run<2, 1>(W, grid, stream)
               : power == 1.0f ? ap_launch_spec_run<1, 1>(W, grid, stream) : ap_launch_spec_run<0, 1>(W, grid, stream);
    return power == 2.0f ? ap_launch_spec_run<2, 0>(W, grid, stream)
           : power == 1.0f ? ap_launch_spec_run<1, 0>(W, grid, stream) : ap_launch_spec_run<0, 0>(W, grid, stream);
}

int ap_frame_stats_f32(const float *y, int64_t B, int64_t L, int frame_length, int hop, int center, int pad_mode,
                       int64_t T, float *rms, float *zcr, void *stream) {
    if (!y) AP_FAIL(AP_ERR_INVALID, "frame statistics: NULL buffer");
    if (frame_length <= 0) AP_FAIL(AP_ERR_INVALID, "frame_length must be positive, got %d", frame_length);
    if (hop <= 0) AP_FAIL(AP_ERR_INVALID, "hop_length must be positive, got %d", hop);
    if (pad_mode != AP_PAD_CONSTANT && pad_mode != AP_PAD_EDGE)
        AP_FAIL(AP_ERR_INVALID, "Unknown pad_mode. Supported: 'constant', 'edge'");
    if (B <= 0 || L <= 0) AP_FAIL(AP_ERR_INVALID, "frame statistics: signal must be non-empty");
    const int pad = center ? frame_length / 2 : 0;
    const int64_t Lp = L + 2 * (int64_t)pad;
    if (Lp < frame_length)
        AP_FAIL(AP_ERR_INVALID, "Signal length (%lld) must be >= frame_length (%d). Consider padding the signal.",
                (long long)Lp, frame_length);
    if (T != 1 + (Lp - frame_length) / hop)
        AP_FAIL(AP_ERR_INVALID, "frame statistics: n_frames mismatch (got %lld)", (long long)T);
    if (!rms && !zcr) return AP_OK;
    // frames per workgroup: the contiguous span (G - 1) hop + frame_length has to fit 64 KiB of LDS
    // frame_length = m hop: every sample read once (block partial sums)
    if (hop % 4 == 0 && frame_length % hop == 0 && frame_length / hop <= 16 && !std::getenv("AP_FRAME_STATS_SPAN")) {
        ApFrameBlocksParams Q;
        Q.y = y; Q.rms = rms; Q.zcr = zcr; Q.L = L; Q.T = T;
        Q.frame_length = frame_length; Q.hop = hop; Q.pad = pad; Q.pad_mode = pad_mode;
        Q.m = frame_length / hop;
        int64_t G = APF_MAX_BLOCKS - Q.m + 1;
        if (G > T) G = T;
        // enough workgroups to fill the chip when the batch is small
        while (G > 16 && ((T + G - 1) / G) * B < 1024) G = (G + 1) / 2;
        Q.G = (int)G;
        Q.tiles_per_clip = (T + G - 1) / G;
        if (Q.tiles_per_clip * B <= kApMaxGrid) {
            hipLaunchKernelGGL(ap_frame_stats_blocks_kernel, dim3((unsigned)(Q.tiles_per_clip * B)), dim3(AP_BLOCK), 0,
                               (hipStream_t)stream, Q);
            return ap_check_launch("ap_frame_stats_f32(blocks)");
        }
    }
    const int64_t budget = 16 * 1024;
    if (frame_length > 36 * 1024) AP_FAIL(AP_ERR_UNSUPPORTED, "frame_length %d does not fit LDS", frame_length);
    int64_t G = frame_length >= budget ? 1 : (budget - frame_length) / hop + 1;
    if (G > 64) G = 64;
    if (G > T) G = T;
    ApFrameStatsParams P;
    P.y = y; P.rms = rms; P.zcr = zcr; P.L = L; P.T = T;
    P.frame_length = frame_length; P.hop = hop; P.pad = pad; P.pad_mode = pad_mode; P.G = (int)G;
    P.tiles_per_clip = (T + G - 1) / G;
    if (P.tiles_per_clip * B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "frame statistics: grid too large");
    const int lds = (int)(((G - 1) * hop + frame_length) * sizeof(float));
    int rc = ap_allow_lds(ap_frame_stats_kernel, lds);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_frame_stats_kernel, dim3((unsigned)(P.tiles_per_clip * B)), dim3(AP_BLOCK), lds,
                       (hipStream_t)stream, P);
    return ap_check_launch("ap_frame_stats_f32");
}

int ap_preemphasis_f32(const float *y, int64_t B, int64_t L, float coef, const float *zi, float *out, float *zf,
                       void *stream) {
    if (!y || !out) AP_FAIL(AP_ERR_INVALID, "preemphasis: NULL buffer");
    if (!(coef >= 0.0f && coef <= 1.0f)) AP_FAIL(AP_ERR_INVALID, "coef must be in [0, 1], got %g", (double)coef);
    if (B <= 0 || L <= 0) AP_FAIL(AP_ERR_INVALID, "preemphasis: signal must be non-empty");
    if (L % 4 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(ap_preemphasis4_kernel, dim3(ap_grid_1d(B * L / 4, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                           (hipStream_t)stream, y, B, L, coef, zi, out, zf);
    else
        hipLaunchKernelGGL(ap_preemphasis_kernel, dim3(ap_grid_1d(B * L, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                           (hipStream_t)stream, y, B, L, coef, zi, out, zf);
    return ap_check_launch("ap_preemphasis_f32");
}

int64_t ap_deemphasis_workspace_floats(int64_t B, int64_t L) {
    if (B <= 0 || L <= 0) return 0;
    const int64_t tile = AP_BLOCK * APD_PER;
    int64_t chunk = 4 * tile;                              // 16 384 samples
    const int64_t n_chunks = (L + chunk - 1) / chunk;
    return n_chunks > 1 ? B * n_chunks : 0;
}

int ap_deemphasis_f32(const float *y, int64_t B, int64_t L, float coef, const float *zi, float *out, float *zf,
                      void *stream) {
    return ap_deemphasis_ws_f32(y, B, L, coef, zi, out, zf, nullptr, stream);
}

int ap_deemphasis_ws_f32(const float *y, int64_t B, int64_t L, float coef, const float *zi, float *out, float *zf,
                         float *ws, void *stream) {
    if (!y || !out) AP_FAIL(AP_ERR_INVALID, "deemphasis: NULL buffer");
    if (!(coef >= 0.0f && coef <= 1.0f)) AP_FAIL(AP_ERR_INVALID, "coef must be in [0, 1], got %g", (double)coef);
    if (B <= 0 || L <= 0) AP_FAIL(AP_ERR_INVALID, "deemphasis: signal must be non-empty");
    const int64_t chunk = 4 * (int64_t)AP_BLOCK * APD_PER;
    const int64_t n_chunks = (L + chunk - 1) / chunk;
    if (ws && n_chunks > 1 && B * n_chunks <= kApMaxGrid) {
        // chunked: end states of all chunks, then every chunk from its composed entering state
        hipLaunchKernelGGL(ap_deemphasis_kernel<1>, dim3((unsigned)(B * n_chunks)), dim3(AP_BLOCK), 0, (hipStream_t)stream,
                           y, L, coef, zi, zi ? 0 : 1, out, zf, chunk, (int)n_chunks, ws);
        int rc = ap_check_launch("ap_deemphasis_f32(carries)");
        if (rc != AP_OK) return rc;
        hipLaunchKernelGGL(ap_deemphasis_kernel<2>, dim3((unsigned)(B * n_chunks)), dim3(AP_BLOCK), 0, (hipStream_t)stream,
                           y, L, coef, zi, zi ? 0 : 1, out, zf, chunk, (int)n_chunks, ws);
        return ap_check_launch("ap_deemphasis_f32(chunks)");
    }
    if (B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "deemphasis: grid too large");
    hipLaunchKernelGGL(ap_deemphasis_kernel<0>, dim3((unsigned)B), dim3(AP_BLOCK), 0, (hipStream_t)stream, y, L, coef, zi,
                       zi ? 0 : 1, out, zf, L, 1, nullptr);
    return ap_check_launch("ap_deemphasis_f32");
}

int ap_savgol_f32(const float *x, int64_t outer, int64_t n, int64_t inner, const float *taps, int width, int mode,
                  float cval, const float *edge, float *out, void *stream) {
    if (!x || !taps || !out) AP_FAIL(AP_ERR_INVALID, "delta: NULL buffer");
    if (width < 3) AP_FAIL(AP_ERR_INVALID, "width must be >= 3, got %d", width);
    if (width % 2 == 0) AP_FAIL(AP_ERR_INVALID, "width must be odd, got %d", width);
    if (mode < AP_SG_INTERP || mode > AP_SG_WRAP) AP_FAIL(AP_ERR_INVALID, "delta: unknown mode");
    if (outer <= 0 || n <= 0 || inner <= 0) AP_FAIL(AP_ERR_INVALID, "delta: empty array");
    if (mode == AP_SG_INTERP && (!edge || width > n))
        AP_FAIL(AP_ERR_INVALID, "when mode='interp', width=%d cannot exceed data.shape[axis]=%lld", width, (long long)n);
    if (inner == 1 && width <= 64 && n < (1 << 30)) {       // contiguous axis: one row chunk per workgroup
        const int64_t chunks = (n + AP_BLOCK * APSG_PER - 1) / (AP_BLOCK * APSG_PER);
        if (outer * chunks <= kApMaxGrid) {
            hipLaunchKernelGGL(ap_savgol_rows_kernel, dim3((unsigned)(outer * chunks)), dim3(AP_BLOCK), 0, (hipStream_t)stream,
                               x, outer, (int)n, (int)chunks, taps, width, mode, cval, edge, out);
            return ap_check_launch("ap_savgol_f32(rows)");
        }
    }
    hipLaunchKernelGGL(ap_savgol_kernel, dim3(ap_grid_1d(outer * n * inner, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, x, outer, n, inner, taps, width, mode, cval, edge, out);
    return ap_check_launch("ap_savgol_f32");
}

int64_t ap_autocorrelation_nfft(int64_t n) {
    int64_t N = 1;
    while (N < 2 * n - 1) N *= 2;
    return N;
}

int ap_autocorrelation_f32(const float *y, int64_t B, int64_t n, int64_t max_lag, int normalize, int center,
                           const float *tw1, const float *tw2, float *ws, float *out, void *stream) {
    if (!y || !out || !ws || !tw1 || !tw2) AP_FAIL(AP_ERR_INVALID, "autocorrelation: NULL buffer");
    if (B <= 0 || n <= 0)
        AP_FAIL(AP_ERR_INVALID, "signal must be 1-dimensional (samples,) or 2-dimensional (batch, samples)");
    if (max_lag <= 0 || max_lag > n) AP_FAIL(AP_ERR_INVALID, "autocorrelation: max_lag must be in [1, n]");
    const int64_t N = ap_autocorrelation_nfft(n);
    int n1, n2;
    if (ap_cfft_split(N, &n1, &n2) != 0)
        AP_FAIL(AP_ERR_UNSUPPORTED, "autocorrelation: n_fft %lld does not split into two on-chip legs", (long long)N);
    if (B > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "autocorrelation: grid too large");
    ap_float2 *bufA = reinterpret_cast<ap_float2 *>(ws);
    ap_float2 *bufB = bufA + B * N;
    float *padded = reinterpret_cast<float *>(bufB);           // consumed by leg 1 before leg 2 overwrites it
    float *mean = ws + 4 * B * N;
    hipStream_t st = (hipStream_t)stream;
    if (center) {
        hipLaunchKernelGGL(ap_row_mean_kernel, dim3((unsigned)B), dim3(AP_BLOCK), 0, st, y, n, mean);
        int rc = ap_check_launch("ap_row_mean");
        if (rc != AP_OK) return rc;
    }
    hipLaunchKernelGGL(ap_autocorr_pad_kernel, dim3(ap_grid_1d(B * N, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0, st,
                       y, B, n, N, center ? mean : nullptr, padded);
    int rc = ap_check_launch("ap_autocorr_pad");
    if (rc != AP_OK) return rc;
    ApCfftParams L1, L2;
    rc = ap_prepare_cfft(L1, L2, padded, bufA, bufB, B, N, n1, n2, tw1, tw2, 0, 1, 0, 1.0f);   // forward, real in
    if (rc != AP_OK) return rc;
    rc = ap_launch_cfft_leg(L1, B, stream);
    if (rc != AP_OK) return rc;
    rc = ap_launch_cfft_leg(L2, B, stream);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_power_spectrum_kernel, dim3(ap_grid_1d(B * N, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0, st,
                       bufB, B * N);
    rc = ap_check_launch("ap_power_spectrum");
    if (rc != AP_OK) return rc;
    float *r = reinterpret_cast<float *>(bufB);                 // inverse leg 2 writes the real rows over leg 1's input array
    rc = ap_prepare_cfft(L1, L2, bufB, bufA, r, B, N, n1, n2, tw1, tw2, 1, 0, 1, (float)(1.0 / (double)N));
    if (rc != AP_OK) return rc;
    rc = ap_launch_cfft_leg(L1, B, stream);
    if (rc != AP_OK) return rc;
    rc = ap_launch_cfft_leg(L2, B, stream);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_autocorr_finish_kernel, dim3(ap_grid_1d(B * max_lag, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK),
                       0, st, r, B, N, max_lag, normalize, out);
    return ap_check_launch("ap_autocorrelation_f32");
}

int ap_spectral_contrast_f32(const float *S, int64_t B, int64_t F, int64_t T, const int32_t *bands_dev, int n_bands,
                             int linear, float *out, void *stream) {
    if (!S || !bands_dev || !out) AP_FAIL(AP_ERR_INVALID, "spectral_contrast: NULL buffer");
    if (B <= 0 || F <= 0 || T <= 0)
        AP_FAIL(AP_ERR_INVALID, "S must be 2D (freq_bins, n_frames) or 3D (batch, freq_bins, n_frames)");
    if (n_bands <= 0 || n_bands > 65535) AP_FAIL(AP_ERR_INVALID, "n_bands must be positive");
    const int64_t blocks = (B * T + AP_BLOCK - 1) / AP_BLOCK;
    if (blocks > kApMaxGrid) AP_FAIL(AP_ERR_UNSUPPORTED, "spectral_contrast: grid too large");
    hipLaunchKernelGGL(ap_spectral_contrast_kernel, dim3((unsigned)blocks, (unsigned)n_bands), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, S, B, F, T, bands_dev, linear, out);
    return ap_check_launch("ap_spectral_contrast_f32");
}

int ap_acf_peaks_f32(const float *r, int64_t rows, int n_lag, int min_lag, int max_lag, float threshold, float sr,
                     float *f0, unsigned char *voiced, float *periodicity, void *stream) {
    if (!r) AP_FAIL(AP_ERR_INVALID, "pitch: NULL buffer");
    if (rows <= 0 || n_lag <= 0) AP_FAIL(AP_ERR_INVALID, "pitch: empty autocorrelation");
    if (min_lag < 0) AP_FAIL(AP_ERR_INVALID, "pitch: negative lag");
    hipLaunchKernelGGL(ap_acf_peak_kernel, dim3((unsigned)((rows + AP_BLOCK - 1) / AP_BLOCK)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, r, rows, n_lag, min_lag, max_lag, threshold, sr, f0, voiced, periodicity);
    return ap_check_launch("ap_acf_peaks_f32");
}

int ap_pcm16_to_f32(const int16_t *x, int64_t n, float scale, float *out, void *stream) {
    if (n < 0 || (n > 0 && (!x || !out))) AP_FAIL(AP_ERR_INVALID, "pcm16_to_f32: bad buffer");
    if (n == 0) return AP_OK;
    hipLaunchKernelGGL(ap_pcm16_to_f32_kernel, dim3(ap_grid_1d((n + 7) / 8, AP_BLOCK, kApStreamGrid)), dim3(AP_BLOCK), 0,
                       (hipStream_t)stream, x, n, scale, out);
    return ap_check_launch("ap_pcm16_to_f32");
}

int ap_melspec_pcm16_fused(int64_t L, int n_fft, int hop, int center, int pad_mode, int n_mels, float power,
                           const int32_t *desc) {
    if (n_fft != 2048 || power != 2.0f || !desc || !(desc[0] & AP_PLAN_PARTS) || (desc[0] & AP_PLAN_FORCE_GENERIC)) return 0;
    if (center && pad_mode != AP_PAD_CONSTANT) return 0;
    if ((L & 1) || (hop & 1)) return 0;                  // a dword holds the even-indexed sample and its successor
    if (n_mels > 128 || desc[12] > 256 || desc[15] > 4) return 0;
    // the bounds ap_prepare_mel_run enforces: past them the caller needs the float32 scratch route
    if (L > (1 << 28) || ap_n_frames(L, n_fft, hop, center) > (1 << 24)) return 0;
    return std::getenv("AP_MEL2048_WAVE") ? 0 : 1;
}

int ap_melspec_pcm16_f32(const int16_t *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                         const float *tw, int center, int pad_mode, int64_t T, const float *fb, const int32_t *plan,
                         const int32_t *desc, int n_mels, float power, float *out, uint32_t *max_key_dev,
                         float *scratch_f32, void *stream) {
    if (!y) AP_FAIL(AP_ERR_INVALID, "melspectrogram: NULL buffer");
    // the fused kernel reads sample PAIRS as dwords: the clip base has to be 4-byte aligned (an int16 view that
    // starts at an odd element is not) - such inputs take the conversion pass
    const bool aligned = (reinterpret_cast<uintptr_t>(y) & 3) == 0;
    if (aligned && ap_melspec_pcm16_fused(L, n_fft, hop, center, pad_mode, n_mels, power, desc)) {
        ApStftParams P;
        int rc = ap_prepare_stft(P, reinterpret_cast<const float *>(y), B, L, n_fft, hop, window, tw, center, pad_mode, T);
        if (rc != AP_OK) return rc;
        rc = ap_prepare_mel(P, fb, plan, desc, n_mels, power, out);
        if (rc != AP_OK) return rc;
        ApMelWaveParams W;
        int grid = 0, n_pass = 0;
        if (ap_prepare_mel_run(W, P, B, plan, desc, APM_WAVES, APW_X_COMPLEX, APM_PARTIAL_OFF, &n_pass, &grid) == AP_OK) {
            if (max_key_dev) {
                hipError_t e = hipMemsetD32Async((hipDeviceptr_t)max_key_dev, (int)kApKeyMinusInf, 1, (hipStream_t)stream);
                if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipMemsetD32Async: %s", hipGetErrorString(e));
                W.max_key = max_key_dev;
            }
            return ap_launch_mel_run<2, 1>(W, n_pass, grid, stream);
        }
    }
    if (!scratch_f32) AP_FAIL(AP_ERR_INVALID, "melspectrogram(int16): this shape needs the float32 scratch buffer");
    int rc = ap_pcm16_to_f32(y, B * L, 1.0f / 32768.0f, scratch_f32, stream);
    if (rc != AP_OK) return rc;
    return ap_melspec_max_f32(scratch_f32, B, L, n_fft, hop, window, tw, center, pad_mode, T, fb, plan, desc, n_mels,
                              power, out, max_key_dev, stream);
}

}  // extern "C"
