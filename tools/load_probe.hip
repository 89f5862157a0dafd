// What do row-segment LOADS from the (B, F, T) complex64 spectrum cost as a function of segment length
// and alignment?  Load-only kernels in the ISTFT kernels' order: a persistent 512-thread workgroup walks
// a contiguous stretch of (clip, group-of-G-frames) and, per group, reads G frames of all F rows.
//   build: hipcc --offload-arch=gfx950 -O3 -o build/load_probe tools/load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: segment [gG, gG+G) as it falls; MODE 1: the aligned window of the row that starts inside the group
template <int G, int MODE>
__global__ __launch_bounds__(512) void load_kernel(const float2* __restrict__ in, float* __restrict__ sink, int B, int F, int T, int ng)
{
    const int tid = threadIdx.x;
    constexpr int RPI = 512 / G;
    const long total = (long)B * ng;
    const long per = (total + gridDim.x - 1) / gridDim.x;
    long s0 = per * blockIdx.x, s1 = s0 + per;
    if (s1 > total) s1 = total;
    const int pos = tid % G, r0 = tid / G;
    float acc = 0.0f;
    for (long s = s0; s < s1; ++s) {
        const int b = (int)(s / ng), g = (int)(s % ng);
#pragma unroll 4
        for (int k = r0; k < F; k += RPI) {
            const long row = (long)b * F + k;
            long t = (long)g * G + pos;
            if (MODE == 1) t = (long)g * G - ((row * T + (long)g * G) % G) + pos;
            if (t >= 0 && t < T) { const float2 v = in[row * T + t]; acc += v.x + v.y; }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int G, int MODE>
static void run(const char* name, const float2* in, float* sink, int B, int F, int T, int grid)
{
    const int ng = (T + G - 1) / G;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) load_kernel<G, MODE><<<grid, 512>>>(in, sink, B, F, T, ng);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) load_kernel<G, MODE><<<grid, 512>>>(in, sink, B, F, T, ng);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms / 50);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)B * F * T * 8;
    printf("%-44s grid %4d  %.4f ms  %.2f TB/s\n", name, grid, ts[2], bytes / ts[2] / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    int B = 256, F = 1025, T = 431;
    if (argc > 3) { B = atoi(argv[1]); F = atoi(argv[2]); T = atoi(argv[3]); }
    const size_t n = (size_t)B * F * T;
    float2* in; float* sink;
    CK(hipMalloc(&in, n * 8 + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 0, n * 8));
    printf("loads: B %d F %d T %d: %.0f MB\n", B, F, T, n * 8 / 1e6);
    for (int i = 0; i < 300; ++i) load_kernel<16, 1><<<512, 512>>>(in, sink, B, F, T, (T + 15) / 16);
    CK(hipDeviceSynchronize());
    for (int grid : {256, 512}) {
        run<8, 0>("G=8  64 B segs, as they fall", in, sink, B, F, T, grid);
        run<8, 1>("G=8  64 B aligned windows", in, sink, B, F, T, grid);
        run<16, 0>("G=16 128 B segs, as they fall", in, sink, B, F, T, grid);
        run<16, 1>("G=16 128 B aligned windows", in, sink, B, F, T, grid);
        run<32, 0>("G=32 256 B segs, as they fall", in, sink, B, F, T, grid);
        run<32, 1>("G=32 256 B aligned windows", in, sink, B, F, T, grid);
    }
    return 0;
}
