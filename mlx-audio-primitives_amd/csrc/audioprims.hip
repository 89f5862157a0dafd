// libaudioprims_hip.so — C ABI entry points (include/audioprims.h): validate
// (ap_launch.h), then enqueue the gfx950 kernels.  Nothing here allocates device
// memory or synchronises.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "ap_launch.h"
#include "kernels_generic.h"
#include "kernels_wave.h"

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

template <int PMODE>
static int ap_launch_mel_wave(const ApMelWaveParams &W, int grid, void *stream) {
    int rc = ap_allow_lds(ap_mel2048_wave_kernel<PMODE>, W.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_mel2048_wave_kernel<PMODE>, dim3(grid), dim3(256), W.lds_bytes,
                       (hipStream_t)stream, W);
    return ap_check_launch("ap_melspec_f32(wave)");
}

extern "C" {

int ap_version(void) { return 100; }

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
    rc = ap_allow_lds(ap_stft_generic_kernel<0>, P.tile.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_stft_generic_kernel<0>, dim3((unsigned)(P.tiles_per_clip * B)),
                       dim3(AP_BLOCK), P.tile.lds_bytes, (hipStream_t)stream, P);
    return ap_check_launch("ap_stft_f32");
}

int ap_melspec_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                   const float *tw, int center, int pad_mode, int64_t T, const float *fb,
                   const int32_t *plan, const int32_t *desc, int n_mels, float power, float *out,
                   void *stream) {
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, n_fft, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    rc = ap_prepare_mel(P, fb, plan, desc, n_mels, power, out);
    if (rc != AP_OK) return rc;
    if (ap_mel_wave_eligible(n_fft, plan, desc)) {
        ApMelWaveParams W;
        int grid = 0;
        if (ap_prepare_mel_wave(W, P, B, plan, desc, &grid) == AP_OK) {
            if (power == 2.0f) return ap_launch_mel_wave<2>(W, grid, stream);
            if (power == 1.0f) return ap_launch_mel_wave<1>(W, grid, stream);
            return ap_launch_mel_wave<0>(W, grid, stream);
        }
    }
    rc = ap_allow_lds(ap_stft_generic_kernel<1>, P.tile.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_stft_generic_kernel<1>, dim3((unsigned)(P.tiles_per_clip * B)),
                       dim3(AP_BLOCK), P.tile.lds_bytes, (hipStream_t)stream, P);
    return ap_check_launch("ap_melspec_f32");
}

int ap_irfft_frames_f32(const float *S, int64_t B, int64_t T, int n_fft, const float *tw,
                        float *frames, void *stream) {
    ApIrfftParams P;
    int rc = ap_prepare_irfft(P, S, B, T, n_fft, tw, frames);
    if (rc != AP_OK) return rc;
    rc = ap_allow_lds(ap_irfft_generic_kernel, P.tile.lds_bytes);
    if (rc != AP_OK) return rc;
    hipLaunchKernelGGL(ap_irfft_generic_kernel, dim3((unsigned)(P.tiles_per_clip * B)),
                       dim3(AP_BLOCK), P.tile.lds_bytes, (hipStream_t)stream, P);
    return ap_check_launch("ap_irfft_frames_f32");
}

int ap_istft_f32(const float *S, int64_t B, int64_t T, int n_fft, int hop, const float *window,
                 const float *tw, float *frames_ws, int64_t out_offset, int64_t out_len, float *out,
                 void *stream) {
    if (!frames_ws) AP_FAIL(AP_ERR_INVALID, "istft: NULL workspace");
    int rc = ap_irfft_frames_f32(S, B, T, n_fft, tw, frames_ws, stream);
    if (rc != AP_OK) return rc;
    return ap_overlap_add_f32(frames_ws, window, B, T, n_fft, hop, out_offset, out_len, out, stream);
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

}  // extern "C"
