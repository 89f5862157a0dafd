// Per-CU cost of the STFT's row stores by width, away from the chip-wide HBM limit: a persistent 512-thread workgroup on
// each of G CUs (G = 256, 64, 16) writes whole 128-byte lines of a (B, 1024, Ts) complex array, 131 KB per group, as
//   W8 : 8-byte stores, a wave instruction = 4 rows x 16 frames (the kernels' pattern: 32 per thread and group)
//   W16: 16-byte stores, a wave instruction = 8 rows x (8 lanes x 2 frames): 16 per thread and group
// with the data ready in registers (nothing else in the loop).
//   build: hipcc --offload-arch=gfx950 -O3 -o build/store_width_probe tools/store_width_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W16>
__global__ __launch_bounds__(512, 2) void probe(float* __restrict__ out, int F, int Ts, int ng, long total)
{
    const int tid = threadIdx.x;
    const long s0 = total * blockIdx.x / gridDim.x, s1 = total * (blockIdx.x + 1) / gridDim.x;
    const f2 a = {(float)tid, 1.0f};
    const f4 q = {(float)tid, 1.0f, 2.0f, 3.0f};
    for (long s = s0; s < s1; ++s) {
        const int b = (int)(s / ng), g = (int)(s % ng);
        char* base = reinterpret_cast<char*>(out) + (long)b * F * Ts * 8;
        if (!W16) {
            char* p = base + ((long)(tid >> 4) * Ts + g * 16 + (tid & 15)) * 8;
#pragma unroll
            for (int i = 0; i < 32; ++i) *reinterpret_cast<f2*>(p + (long)(32 * i) * Ts * 8) = a;
        } else {
            char* p = base + ((long)(tid >> 3) * Ts + g * 16 + 2 * (tid & 7)) * 8;
#pragma unroll
            for (int i = 0; i < 16; ++i) *reinterpret_cast<f4*>(p + (long)(64 * i) * Ts * 8) = q;
        }
    }
}

template <int W16>
static void run(const char* name, float* out, int B, int F, int Ts, int grid)
{
    const int ng = Ts / 16;
    const long total = (long)(B * grid / 256) * ng;            // the same work per workgroup at every grid size
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<W16>, dim3(grid), dim3(512), 0, 0, out, F, Ts, ng, total);
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<W16>, dim3(grid), dim3(512), 0, 0, out, F, Ts, ng, total);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t / 20);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)total * 1024 * 16 * 8;
    printf("%-34s grid %3d  %.4f ms  %.2f TB/s  = %.1f B/clk per CU at 2.4 GHz\n", name, grid, ms[2], bytes / ms[2] / 1e9, bytes / grid / (ms[2] * 1e-3) / 2.4e9);
}

int main()
{
    const int B = 256, F = 1024, Ts = 432;
    float* out;
    CK(hipMalloc(&out, (size_t)B * F * Ts * 8));
    for (int grid : {256, 64, 16}) {
        run<0>("8-byte stores, 4 rows x 16 frames", out, B, F, Ts, grid);
        run<1>("16-byte stores, 8 rows x 16 frames", out, B, F, Ts, grid);
    }
    return 0;
}
