// Translation unit of the fused n_fft = 2048 ISTFT with 16-frame loads (kernels_istft16.h).
#include <hip/hip_runtime.h>
#include <cstdlib>

#define AP_TU_SECONDARY 1
#include "ap_tu.h"
#include "kernels_istft16.h"

int ap_launch_istft16(const float *S, const float *tw, int64_t B, int64_t T, int64_t Ts, const float *window, int hop,
                      int64_t out_offset, int64_t out_len, float *out, void *stream) {
    ApIstft16Params W;
    int grid = 0;
    if (ap_prepare_istft16(W, S, tw, B, T, Ts, window, hop, out_offset, out_len, out, &grid) != AP_OK) return 1;
    if (ap_g16_grid_cap > 0 && grid > ap_g16_grid_cap) grid = ap_g16_grid_cap;
    auto kern = hop == 256 ? ap_istft2048_g16_kernel<8> : hop == 512 ? ap_istft2048_g16_kernel<9> : ap_istft2048_g16_kernel<10>;
    static const bool loose = std::getenv("AP_ISTFT16_LOOSE") != nullptr;            // A/B switch: the unfenced transform
    if (loose && hop == 512) kern = ap_istft2048_g16_kernel<9, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       W.lds_bytes);
    if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipFuncSetAttribute(LDS=%d): %s", W.lds_bytes, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APS_WAVES), W.lds_bytes, (hipStream_t)stream, W);
    e = hipGetLastError();
    if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "ap_istft_f32(g16): %s", hipGetErrorString(e));
    return AP_OK;
}
