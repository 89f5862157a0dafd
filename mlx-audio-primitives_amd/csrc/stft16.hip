// Translation unit of the 16-frames-per-group STFT kernel (kernels_stft16.h).
#include <hip/hip_runtime.h>
#include <cstdlib>

#define AP_TU_SECONDARY 1
#include "ap_tu.h"
#include "kernels_stft16.h"

thread_local int ap_g16_grid_cap = 0;

template <int PADGEN, int ALIGNED, int NT, int GL = 0, int T2 = 0>
static int ap_stft16_go(const ApStft16Params &W, int grid, void *stream) {
    if (ap_g16_grid_cap > 0 && grid > ap_g16_grid_cap) grid = ap_g16_grid_cap;
#ifdef AP_PHASE_CLOCK
    static const int diag_cap = std::getenv("AP_G16_GRID_CAP") ? std::atoi(std::getenv("AP_G16_GRID_CAP")) : 0;   // diagnostic build only
    if (diag_cap > 0 && grid > diag_cap) grid = diag_cap;
#endif
    auto kern = ap_stft2048_g16_kernel<PADGEN, ALIGNED, NT, GL, T2>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, W.lds_bytes);
    if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "hipFuncSetAttribute(LDS=%d): %s", W.lds_bytes, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * APS_WAVES), W.lds_bytes, (hipStream_t)stream, W);
    e = hipGetLastError();
    if (e != hipSuccess) AP_FAIL(AP_ERR_HIP, "ap_stft_f32(g16): %s", hipGetErrorString(e));
    return AP_OK;
}

int ap_launch_stft16(const ApStftParams &P, int64_t B, int64_t Ts, void *stream) {
    ApStft16Params W;
    int grid = 0, aligned = 0;
    if (ap_prepare_stft16(W, P, B, Ts, &grid, &aligned) != AP_OK) return 1;
    // A/B switches for measurements: AP_STFT16_NT=1 non-temporal stores, AP_STFT16_CARRY=1 keeps the carry path
    // on an aligned layout
    static const bool nt = std::getenv("AP_STFT16_NT") && std::atoi(std::getenv("AP_STFT16_NT")) != 0;
    static const bool force_carry = std::getenv("AP_STFT16_CARRY") != nullptr;
    if (force_carry) aligned = 0;
    static const int stagger = std::getenv("AP_STFT16_STAGGER") ? std::atoi(std::getenv("AP_STFT16_STAGGER")) : 0;
    W.stagger = stagger;
    const bool pg = !ap_clip_loads_ok(W);
    // A/B switch: AP_STFT16_T2=1 the whole-group tile (kernels_stft16.h; measured slower, not the default)
    static const int t2 = std::getenv("AP_STFT16_T2") ? std::atoi(std::getenv("AP_STFT16_T2")) : 0;
    if (aligned && t2 == 1 && Ts <= 490000) return pg   // (a clip is one buffer resource there: 1025 Ts 8 < 0xF0000000 bytes)
        ? ap_stft16_go<1, 1, 0, 0, 1>(W, grid, stream) : ap_stft16_go<0, 1, 0, 0, 1>(W, grid, stream);
    if (aligned) {
        if (pg) return nt ? ap_stft16_go<1, 1, 1>(W, grid, stream) : ap_stft16_go<1, 1, 0>(W, grid, stream);
        return nt ? ap_stft16_go<0, 1, 1>(W, grid, stream) : ap_stft16_go<0, 1, 0>(W, grid, stream);
    }
    if (pg) return nt ? ap_stft16_go<1, 0, 1>(W, grid, stream) : ap_stft16_go<1, 0, 0>(W, grid, stream);
    return nt ? ap_stft16_go<0, 0, 1>(W, grid, stream) : ap_stft16_go<0, 0, 0>(W, grid, stream);
}

// STFT of y with the Griffin-Lim projection in its store phase (kernels_stft16.h, GL = 1): raw spectrum -> P.out_c,
// new estimate -> rebuilt; both and `prev` in rows Ts apart (Ts % 16 == 0, 128-byte aligned), `mag` dense.
int ap_launch_stft16_gl(const ApStftParams &P, int64_t B, int64_t Ts, const float *prev, const float *mag, float momentum,
                        float *rebuilt, void *stream) {
    ApStft16Params W;
    int grid = 0, aligned = 0;
    if (ap_prepare_stft16(W, P, B, Ts, &grid, &aligned) != AP_OK) return 1;
    if (!aligned || ((reinterpret_cast<uintptr_t>(prev) | reinterpret_cast<uintptr_t>(rebuilt)) & 127)) return 1;
    W.gl_prev = reinterpret_cast<const ap_float2 *>(prev);
    W.gl_mag = mag;
    W.gl_rebuilt = reinterpret_cast<ap_float2 *>(rebuilt);
    W.gl_momentum = momentum;
    // A/B switch: AP_GL_PAIRS=1 two frames per thread as 16-byte accesses (GL = 2): half the vector-memory instructions,
    // bit-identical results, no faster (cfg3, 32 iterations, same box: 5.31-5.35 ms against 5.23-5.27; one stream 5.90 / 5.92)
    static const bool pairs = std::getenv("AP_GL_PAIRS") && std::atoi(std::getenv("AP_GL_PAIRS")) != 0;
    if (pairs) return ap_clip_loads_ok(W) ? ap_stft16_go<0, 1, 0, 2>(W, grid, stream) : ap_stft16_go<1, 1, 0, 2>(W, grid, stream);
    if (!ap_clip_loads_ok(W)) return ap_stft16_go<1, 1, 0, 1>(W, grid, stream);
    return ap_stft16_go<0, 1, 0, 1>(W, grid, stream);
}

#ifdef AP_PHASE_CLOCK
// diagnostic build only (tools/phase_clock.py)
extern "C" int ap_phase_read_stft16(unsigned long long *host, int n_words) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(ap_phase_clk), sizeof(unsigned long long) * (size_t)n_words);
    return e == hipSuccess ? 0 : 1;
}
#endif
