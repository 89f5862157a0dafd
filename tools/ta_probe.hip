// What does ISSUING the ISTFT's spectrum loads cost a CU, and does the width of a load matter?  One persistent
// 512-thread workgroup per CU walks its stretch of (clip, 16-frame group) like ap_istft2048_g16_kernel and, per
// group, loads 16 frames of all 1024 rows (bin 512 left out) with one group in flight (registers), in the
// kernel's pattern or in wider ones.  Wave 0 clocks the issue of each group's loads (s_memtime; no wait for data).
//   MODE 0: 8-byte loads, a wave instruction = 4 rows x 16 frames (the kernel's: 32 loads per thread and group)
//   MODE 1: 16-byte loads, a wave instruction = 8 rows x 16 frames (8 lanes x 2 frames per row): 16 loads
//   MODE 2: 8-byte loads issued in 8 bursts of 4 separated by ~500 cycles of arithmetic (spread)
//   build: hipcc --offload-arch=gfx950 -O3 -o build/ta_probe tools/ta_probe.hip      run: build/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_clk[256 * 2];

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const float* __restrict__ S, float* __restrict__ sink, int B, int F, int Ts, int ng, int spin)
{
    const int tid = threadIdx.x;
    const long total = (long)B * ng;
    const long s0 = total * blockIdx.x / gridDim.x, s1 = total * (blockIdx.x + 1) / gridDim.x;
    float acc = 0.0f;
    unsigned long long t_issue = 0, t_all0, t_all1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_all0)::"memory");
    f2 a[32];
    f4 q[16];
    auto issue = [&](long s, int c0, int c1) {
        const int b = (int)(s / ng), g = (int)(s % ng);
        const char* base = reinterpret_cast<const char*>(S) + (long)b * F * Ts * 8;
        if (MODE != 1) {
            const int sq = tid >> 4, sf = tid & 15;
            const char* p = base + ((long)sq * Ts + g * 16 + sf) * 8;
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (i >= c0 && i < c1) a[i] = *reinterpret_cast<const f2*>(p + (long)(32 * i) * Ts * 8);
        } else {
            const int sq = tid >> 3, sp = tid & 7;
            const char* p = base + ((long)sq * Ts + g * 16 + 2 * sp) * 8;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i >= c0 && i < c1) q[i] = *reinterpret_cast<const f4*>(p + (long)(64 * i) * Ts * 8);
        }
    };
    auto clocked = [&](long s, int c0, int c1) {
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        issue(s, c0, c1);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        t_issue += t1 - t0;
    };
    for (long s = s0; s < s1; ++s) {
        __builtin_amdgcn_s_barrier();
        if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                clocked(s, 4 * k, 4 * k + 4);
                float x = acc;                              // ~spin dependent FMAs: the "transform"
                for (int j = 0; j < spin; ++j) x = __builtin_fmaf(x, 1.0001f, 0.5f);
                acc = x;
            }
        } else {
            clocked(s, 0, 32);
            float x = acc;
            for (int j = 0; j < 8 * spin; ++j) x = __builtin_fmaf(x, 1.0001f, 0.5f);
            acc = x;
        }
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc += a[i].x + a[i].y;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += q[i].x + q[i].y + q[i].z + q[i].w;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_all1)::"memory");
    if (tid == 0) { g_clk[2 * blockIdx.x] = t_issue; g_clk[2 * blockIdx.x + 1] = t_all1 - t_all0; }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, const float* S, float* sink, int B, int F, int Ts, int spin)
{
    const int ng = Ts / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, S, sink, B, F, Ts, ng, spin);
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, S, sink, B, F, Ts, ng, spin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t / 20);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<unsigned long long> clk(512);
    CK(hipMemcpyFromSymbol(clk.data(), HIP_SYMBOL(g_clk), sizeof(unsigned long long) * 512));
    std::vector<double> is, al;
    for (int i = 0; i < 256; ++i) { is.push_back((double)clk[2 * i]); al.push_back((double)clk[2 * i + 1]); }
    std::sort(is.begin(), is.end()); std::sort(al.begin(), al.end());
    const double groups = (double)B * ng / 256.0;
    const double bytes = (double)B * 1024 * Ts * 8;
    printf("%-44s spin %5d  %.4f ms  %.2f TB/s   issue %.0f cycles/group (%.1f %% of %.0f)\n", name, spin, ms[2], bytes / ms[2] / 1e9,
           is[128] / groups, 100.0 * is[128] / al[128], al[128] / groups);
}

int main()
{
    const int B = 256, F = 1025, Ts = 432;
    float *S, *sink;
    CK(hipMalloc(&S, (size_t)B * F * Ts * 8)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(S, 0, (size_t)B * F * Ts * 8));
    for (int spin : {0, 500, 2000, 4000}) {
        run<0>("8-byte loads, 4 rows x 16 frames, one burst", S, sink, B, F, Ts, spin);
        run<1>("16-byte loads, 8 rows x 16 frames, one burst", S, sink, B, F, Ts, spin);
        run<2>("8-byte loads, 8 bursts of 4 between arithmetic", S, sink, B, F, Ts, spin);
    }
    return 0;
}
