// What do row-segment stores into the (B, F, T) complex64 spectrum cost as a function of the segment
// length and its alignment?  Store-only kernels in the STFT kernels' emission order: a persistent
// 512-thread workgroup walks a contiguous stretch of (clip, group-of-G-frames) and, per group, writes
// G frames of all F rows of the clip (row stride T * 8 bytes, T odd = every alignment occurs).
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/store_probe.hip
//   run:   /tmp/store_probe [B F T]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// G frames per group; V complex values per lane and store (1 = 8 bytes, 2 = 16 bytes);
// MODE 0: segment [gG, gG+G) as it falls (misaligned rows straddle lines)
// MODE 1: every row's window shifted back to the previous G*8-byte boundary of ITS row (what the
//         register-carry scheme emits: whole aligned windows, clipped to the row)
// MODE 2: segment [gG, gG+G) cut at the row's line boundary into two stores issued by different lanes
//         (same bytes as MODE 0; shows whether the hardware's own split of a straddling access matters)
// NT: nontemporal stores
template <int G, int V, int MODE, int NT>
__global__ __launch_bounds__(512) void store_kernel(float2* __restrict__ out, int B, int F, int T, int n_groups_per_clip)
{
    const int tid = threadIdx.x;
    constexpr int LPR = G / V;                // lanes per row segment
    constexpr int RPI = 512 / LPR;            // rows per iteration of the workgroup
    const long total = (long)B * n_groups_per_clip;
    const long per = (total + gridDim.x - 1) / gridDim.x;
    long s0 = per * blockIdx.x, s1 = s0 + per;
    if (s1 > total) s1 = total;
    const int pos = (tid % LPR) * V;
    const int r0 = tid / LPR;
    for (long s = s0; s < s1; ++s) {
        const int b = (int)(s / n_groups_per_clip), g = (int)(s % n_groups_per_clip);
        for (int k = r0; k < F; k += RPI) {
            const long row = (long)b * F + k;
            long t = (long)g * G + pos;
            if (MODE == 1) {
                // window start = largest multiple-of-G element index (in the flat array) <= row*T + gG
                const long flat = row * T + (long)g * G;
                t = (long)g * G - (flat % G) + pos;
                // (the last, partial window of a row is emitted by the group after the last: skip here, the
                //  byte count per launch differs by < 1 %)
            }
            float2 v = make_float2((float)tid, (float)g);
            float2* p = out + row * T + t;
            if (V == 1) {
                if (t >= 0 && t < T) {
                    if (NT) { vf2 q = {v.x, v.y}; __builtin_nontemporal_store(q, (vf2*)p); } else *p = v;
                }
            } else {
                if (t >= 0 && t + 1 < T) {
                    float4 w = make_float4(v.x, v.y, v.x, v.y);
                    if ((((size_t)p) & 15) == 0) {
                        if (NT) { vf4 q = {w.x, w.y, w.z, w.w}; __builtin_nontemporal_store(q, (vf4*)p); } else *(float4*)p = w;
                    } else {
                        p[0] = v; p[1] = v;
                    }
                } else if (t >= 0 && t < T) {
                    *p = v;
                }
            }
        }
    }
}

// plain fill for the ceiling
__global__ __launch_bounds__(512) void fill_kernel(float4* __restrict__ out, long n4)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    float4 w = make_float4(1.f, 2.f, 3.f, 4.f);
    for (; i < n4; i += stride) out[i] = w;
}

template <int G, int V, int MODE, int NT>
static void run(const char* name, float2* out, int B, int F, int T, int grid)
{
    const int ng = (T + G - 1) / G + (MODE == 1 ? 0 : 0);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) store_kernel<G, V, MODE, NT><<<grid, 512>>>(out, B, F, T, ng);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) store_kernel<G, V, MODE, NT><<<grid, 512>>>(out, B, F, T, ng);
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
    float2* out;
    CK(hipMalloc(&out, n * 8 + 4096));
    printf("B %d F %d T %d: %.0f MB\n", B, F, T, n * 8 / 1e6);
    // ramp the clocks
    for (int i = 0; i < 2000; ++i) fill_kernel<<<2048, 512>>>((float4*)out, n / 2);
    CK(hipDeviceSynchronize());
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) fill_kernel<<<2048, 512>>>((float4*)out, n / 2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s            %.4f ms  %.2f TB/s\n", "fill (16 B per lane, contiguous)", ms / 50, n * 8 / (ms / 50) / 1e9);
    }
    for (int grid : {256, 512, 1024}) {
        run<8, 1, 0, 0>("G=8  64 B segs, as they fall", out, B, F, T, grid);
        run<8, 1, 1, 0>("G=8  64 B aligned windows", out, B, F, T, grid);
        run<16, 1, 0, 0>("G=16 128 B segs, as they fall", out, B, F, T, grid);
        run<16, 1, 1, 0>("G=16 128 B aligned windows, 8 B/lane", out, B, F, T, grid);
        run<16, 2, 1, 0>("G=16 128 B aligned windows, 16 B/lane", out, B, F, T, grid);
        run<32, 1, 0, 0>("G=32 256 B segs, as they fall", out, B, F, T, grid);
        run<32, 1, 1, 0>("G=32 256 B aligned windows, 8 B/lane", out, B, F, T, grid);
        run<32, 2, 1, 0>("G=32 256 B aligned windows, 16 B/lane", out, B, F, T, grid);
        run<64, 1, 0, 0>("G=64 512 B segs, as they fall", out, B, F, T, grid);
        run<8, 1, 1, 1>("G=8  64 B aligned windows, nt", out, B, F, T, grid);
        run<16, 1, 1, 1>("G=16 128 B aligned windows, nt", out, B, F, T, grid);
        run<16, 1, 0, 1>("G=16 128 B as they fall, nt", out, B, F, T, grid);
    }
    CK(hipFree(out));
    return 0;
}
