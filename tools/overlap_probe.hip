// Do a CU's global stores drain while its waves compute?  One persistent 512-thread workgroup per CU, the STFT's
// store pattern (8-byte stores, a wave instruction = 4 rows x 16 frames = 4 whole 128-byte lines, 131 KB per
// "group" in 8 bursts of 4 stores per thread), and between the bursts either nothing, a chain of VALU FMAs, or LDS
// traffic (ds_write_b64 + ds_read_b64 through a padded per-wave buffer, the transform's kind).  Times: stores
// alone, work alone, both.  If the sum shows up instead of the maximum the two do not overlap.
//   build: hipcc --offload-arch=gfx950 -O3 -o build/overlap_probe tools/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

// WORK 0 none, 1 VALU, 2 LDS;  STORES 0/1
template <int WORK, int STORES>
__global__ __launch_bounds__(512, 2) void probe(float* __restrict__ out, float* __restrict__ sink, int B, int F, int Ts, int ng, int spin)
{
    __shared__ f2 lds[8 * 1100];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f2* X = lds + wave * 1100;
    const long total = (long)B * ng;
    const long s0 = total * blockIdx.x / gridDim.x, s1 = total * (blockIdx.x + 1) / gridDim.x;
    const int sq = tid >> 4, sf = tid & 15;
    f2 acc = {(float)tid, 1.0f};
    for (long s = s0; s < s1; ++s) {
        const int b = (int)(s / ng), g = (int)(s % ng);
        char* p = reinterpret_cast<char*>(out) + (((long)b * F + sq) * Ts + g * 16 + sf) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (STORES) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f2*>(p + (long)(32 * (4 * k + i)) * Ts * 8) = acc;
            }
            if (WORK == 1) {
                f2 x = acc;
                for (int j = 0; j < spin; ++j) x = __builtin_elementwise_fma(x, (f2){1.0001f, 0.9999f}, (f2){0.5f, 0.25f});
                acc = x;
            } else if (WORK == 2) {
                f2 x = acc;
                for (int j = 0; j < spin / 8; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) X[lane * 17 + q] = x + (float)q;
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q = 0; q < 4; ++q) x += X[(lane ^ 5) * 17 + q];
                    __builtin_amdgcn_wave_barrier();
                }
                acc = x * 1e-3f;
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (acc.x == 123.456f) sink[0] = acc.y;
}

template <int WORK, int STORES>
static float run(float* out, float* sink, int B, int F, int Ts, int spin)
{
    const int ng = Ts / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<WORK, STORES>), dim3(256), dim3(512), 0, 0, out, sink, B, F, Ts, ng, spin);
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<WORK, STORES>), dim3(256), dim3(512), 0, 0, out, sink, B, F, Ts, ng, spin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t / 20);
    }
    std::sort(ms.begin(), ms.end());
    return ms[2];
}

int main()
{
    const int B = 256, F = 1024, Ts = 432;
    float *out, *sink;
    CK(hipMalloc(&out, (size_t)B * F * Ts * 8)); CK(hipMalloc(&sink, 64));
    const float st = run<0, 1>(out, sink, B, F, Ts, 0);
    printf("stores alone (%.0f MB)                 %.4f ms  %.2f TB/s\n", (double)B * F * Ts * 8 / 1e6, st, (double)B * F * Ts * 8 / st / 1e9);
    for (int spin : {200, 400, 800}) {
        const float v = run<1, 0>(out, sink, B, F, Ts, spin), vs = run<1, 1>(out, sink, B, F, Ts, spin);
        printf("VALU spin %4d: alone %.4f ms, with stores %.4f ms  (sum %.4f, max %.4f)\n", spin, v, vs, v + st, v > st ? v : st);
        const float l = run<2, 0>(out, sink, B, F, Ts, spin), ls = run<2, 1>(out, sink, B, F, Ts, spin);
        printf("LDS  spin %4d: alone %.4f ms, with stores %.4f ms  (sum %.4f, max %.4f)\n", spin, l, ls, l + st, l > st ? l : st);
    }
    return 0;
}
