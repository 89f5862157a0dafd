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
    // A/B switches: AP_ISTFT16_LOOSE=1 the unfenced transform; AP_ISTFT16_SPREAD=0 the next group's loads issued in the
    // staging pass instead of inside the first step's transform; AP_ISTFT16_TILE=0 the eight-round staging pass
    // (rows / dense layout, round 3: SPREAD 0 TILE 0 0.365 / 0.427 ms, SPREAD 1 TILE 0 0.338 / 0.379, SPREAD 1 TILE 1 0.336 / 0.375)
    static const bool loose = std::getenv("AP_ISTFT16_LOOSE") != nullptr;
    static const int spread = std::getenv("AP_ISTFT16_SPREAD") ? std::atoi(std::getenv("AP_ISTFT16_SPREAD")) : 1;
    static const int tile = std::getenv("AP_ISTFT16_TILE") ? std::atoi(std::getenv("AP_ISTFT16_TILE")) : 1;
#define AP_ISTFT16_PICK(SP, TL) (hop == 256 ? ap_istft2048_g16_kernel<8, true, SP, TL> : hop == 512 ? ap_istft2048_g16_kernel<9, true, SP, TL> \
                                                                                      : ap_istft2048_g16_kernel<10, true, SP, TL>)
    auto kern = tile ? (spread ? AP_ISTFT16_PICK(1, 1) : AP_ISTFT16_PICK(0, 1)) : (spread ? AP_ISTFT16_PICK(1, 0) : AP_ISTFT16_PICK(0, 0));
#undef AP_ISTFT16_PICK
    if (loose && hop == 512) kern = ap_istft2048_g16_kernel<9, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       W.lds_bytes);
    if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipFuncSetAttribute(LDS=%d): %s", W.lds_bytes, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APS_WAVES), W.lds_bytes, (hipStream_t)stream, W);
    e = hipGetLastError();
    if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "ap_istft_f32(g16): %s", hipGetErrorString(e));
    return AP_OK;
}

#ifdef AP_PHASE_CLOCK
// diagnostic build only (tools/phase_clock.py)
extern "C" int ap_phase_read_istft16(unsigned long long *host, int n_words) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(ap_phase_clk), sizeof(unsigned long long) * (size_t)n_words);
    return e == hipSuccess ? 0 : 1;
}
#endif
