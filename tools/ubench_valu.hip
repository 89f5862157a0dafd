// Instruction-throughput microbenchmark for gfx950 (standalone: hipcc --offload-arch=gfx950 -O3
// tools/ubench_valu.hip -o ubench_valu).  It settles the cost model the wave kernels are tuned
// against: how many cycles a SIMD needs per packed-f32 / plain-f32 / DPP / v_cndmask instruction
// at 1-4 waves per SIMD, and what clock the chip holds while every CU runs them.
//
// Every wave runs ITER x 64 independent instructions of one kind (16 destination registers round
// robin, so a dependent instruction is 16 issues away) and stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around the loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int KIND>
__global__ void __launch_bounds__(1024) k_valu(float *out, unsigned long long *stamps, int iters, float seed) {
    f2 a[16], b, c;
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return __builtin_bit_cast(float, 0x3f000000u | (h >> 9)) * ((h & 256) ? 1.0f : -1.0f); };
    b = f2{rnd(), rnd()};
    c = f2{rnd() * seed, rnd()};
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f2{rnd(), rnd()};
    float sc = seed;
    const unsigned long long smask = 0x8888888888888888ull;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#define ONE(i)                                                                                                   \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));                  \
    else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                \
    else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                            \
    else if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                            \
    else if (KIND == 4) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(a[(i + 8) & 15].y)); \
    else if (KIND == 5) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(a[(i + 8) & 15].y), "v"(c.x)); \
    else if (KIND == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));                           \
    else if (KIND == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[i].x) : "v"(a[(i + 8) & 15].y), "v"(b.x) : ); \
    else if (KIND == 8) asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(a[(i + 8) & 15].y), "v"(b.x)); \
    else if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c)); \
    else if (KIND == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.y));                           \
    else if (KIND == 12) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(a[i].x) : "v"(a[(i + 8) & 15].y), "v"(b.x), "s"(smask)); \
    else if (KIND == 13) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(a[(i + 8) & 15].y)); \
    else if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(a[(i + 8) & 15].y)); \
    else if (KIND == 15) asm volatile("v_cndmask_b32_e64 %0, %1, -%2, %3" : "+v"(a[i].x) : "v"(a[(i + 8) & 15].y), "v"(b.x), "s"(smask)); \
    else if (KIND == 16) asm volatile("v_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[i].y)); \
    else if (KIND == 11) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.y), "v"(c.y)); }
            REP16(ONE)
#undef ONE
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float acc = sc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

// LDS kinds: 0 ds_read_b64, 1 ds_read_b128, 2 ds_write_b64, 3 ds_write_b32, 4 ds_read_b32, 5 ds_bpermute_b32
template <int KIND>
__global__ void __launch_bounds__(1024) k_lds(float *out, unsigned long long *stamps, int iters, float seed) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = __builtin_bit_cast(float, 0x3f000000u | (((i + blockIdx.x * 7919u) * 2654435761u) >> 9)) * seed;
    __syncthreads();
    const unsigned base = (unsigned)(wave * 1024 * 4 + lane * (KIND == 1 ? 16 : (KIND == 0 || KIND == 2 ? 8 : 4)));
    float r[4] = {seed, seed, seed, seed};
    f2 v2 = f2{seed, seed};
    float acc = 0.0f;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep) {
            if (KIND == 0) asm volatile("ds_read_b64 %0, %1" : "=v"(v2) : "v"(base) : "memory");
            else if (KIND == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) float *>(r)) : "v"(base) : "memory");
            else if (KIND == 2) asm volatile("ds_write_b64 %0, %1" ::"v"(base), "v"(v2) : "memory");
            else if (KIND == 3) asm volatile("ds_write_b32 %0, %1" ::"v"(base), "v"(r[0]) : "memory");
            else if (KIND == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(r[0]) : "v"(base) : "memory");
            else if (KIND == 5) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r[0]) : "v"(base), "v"(r[1]) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    acc += r[0] + r[1] + r[2] + r[3] + v2.x + v2.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + wave;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

static const size_t LDSB = 100 * 1024;   // > half of the CU's 160 KiB: one workgroup per CU

template <class K>
static void run(const char *name, K kernel, int waves_per_simd, int instr_per_iter, int iters, size_t lds_bytes,
                float *out, unsigned long long *stamps_d) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    const int n_waves = blocks * threads / 64;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, 0, out, stamps_d, iters, 1.0f);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, 0, out, stamps_d, iters, 1.0f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * n_waves);
    CHECK(hipMemcpy(st.data(), stamps_d, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc(n_waves), ghz(n_waves);
    for (int i = 0; i < n_waves; ++i) { cyc[i] = (double)st[2 * i]; ghz[i] = st[2 * i + 1] ? (double)st[2 * i] / ((double)st[2 * i + 1] * 10.0) : 0.0; }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    const double n_instr = (double)instr_per_iter * iters;
    const double med = cyc[n_waves / 2];
    const double agg = ms * 1e-3 * ghz[n_waves / 2] * 1e9 / (n_instr * waves_per_simd);
    std::printf("%-26s waves/SIMD %d  per-wave cyc/instr %6.2f (/waves %5.2f)  kernel-time cyc/instr/SIMD %5.2f  clock %.2f GHz  %.3f ms\n",
                name, waves_per_simd, med / n_instr, med / n_instr / waves_per_simd, agg, ghz[n_waves / 2], ms);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

template <class K>
static void sustain(const char *name, K kernel, int wps, int instr_per_iter, double seconds, float *out, unsigned long long *stamps_d) {
    const int threads = 256 * wps, blocks = 256, iters = 20000;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    int launches = 0;
    float ms = 0;
    do {
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), LDSB, 0, out, stamps_d, iters, 1.0f);
        launches += 10;
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    } while (ms < seconds * 1e3);
    std::vector<unsigned long long> st(2 * blocks * threads / 64);
    CHECK(hipMemcpy(st.data(), stamps_d, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (size_t i = 0; i < st.size() / 2; ++i) if (st[2 * i + 1]) ghz.push_back((double)st[2 * i] / ((double)st[2 * i + 1] * 10.0));
    std::sort(ghz.begin(), ghz.end());
    const double winstr = (double)launches * iters * instr_per_iter * blocks * (threads / 64);
    std::printf("%s waves/SIMD %d: %.3e wave-instructions in %.1f ms = %.3e per second, clock %.3f GHz\n", name, wps, winstr, ms,
                winstr / (ms * 1e-3), ghz[ghz.size() / 2]);
}

int main(int argc, char **argv) {
    if (argc >= 6 && std::string(argv[1]) == "power") {
        float *out;
        unsigned long long *stamps;
        CHECK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
        CHECK(hipMalloc(&stamps, 2 * 256 * 16 * sizeof(unsigned long long)));
        const std::string what = argv[2];
        const int kind = std::atoi(argv[3]), wps = std::atoi(argv[4]);
        const double sec = std::atof(argv[5]);
        if (what == "valu") {
            if (kind == 0) sustain("v_fma_f32", k_valu<0>, wps, 64, sec, out, stamps);
            else if (kind == 1) sustain("v_pk_fma_f32", k_valu<1>, wps, 64, sec, out, stamps);
            else if (kind == 2) sustain("v_pk_add_f32", k_valu<2>, wps, 64, sec, out, stamps);
            else if (kind == 3) sustain("v_pk_mul_f32", k_valu<3>, wps, 64, sec, out, stamps);
            else if (kind == 4) sustain("v_mov_b32_dpp", k_valu<4>, wps, 64, sec, out, stamps);
            else if (kind == 5) sustain("v_fmac_f32_dpp", k_valu<5>, wps, 64, sec, out, stamps);
            else if (kind == 6) sustain("v_add_f32", k_valu<6>, wps, 64, sec, out, stamps);
            else if (kind == 14) sustain("v_mov_b32", k_valu<14>, wps, 64, sec, out, stamps);
            else if (kind == 11) sustain("2x v_fma_f32", k_valu<11>, wps, 128, sec, out, stamps);
        } else {
            if (kind == 0) sustain("ds_read_b64", k_lds<0>, wps, 16, sec, out, stamps);
            else if (kind == 1) sustain("ds_read_b128", k_lds<1>, wps, 16, sec, out, stamps);
            else if (kind == 2) sustain("ds_write_b64", k_lds<2>, wps, 16, sec, out, stamps);
            else if (kind == 3) sustain("ds_write_b32", k_lds<3>, wps, 16, sec, out, stamps);
            else if (kind == 4) sustain("ds_read_b32", k_lds<4>, wps, 16, sec, out, stamps);
        }
        return 0;
    }
    float *out;
    unsigned long long *stamps;
    CHECK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
    CHECK(hipMalloc(&stamps, 2 * 256 * 16 * sizeof(unsigned long long)));
    const int iters = 2000;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_mov_b32_dpp", "v_fmac_f32_dpp",
                           "v_add_f32", "v_cndmask_b32 vcc", "v_add_f32_dpp", "v_pk_fma_f32 op_sel/neg", "v_mul_f32", "2x v_fma_f32 (x,y)",
                           "v_cndmask_b32_e64 sgpr", "v_bfi_b32", "v_mov_b32", "v_cndmask_e64 neg", "v_swap_b32"};
    for (int wps = 2; wps <= 3; ++wps) {
        run(names[0], k_valu<0>, wps, 64, iters, LDSB, out, stamps);
        run(names[1], k_valu<1>, wps, 64, iters, LDSB, out, stamps);
        run(names[2], k_valu<2>, wps, 64, iters, LDSB, out, stamps);
        run(names[3], k_valu<3>, wps, 64, iters, LDSB, out, stamps);
        run(names[4], k_valu<4>, wps, 64, iters, LDSB, out, stamps);
        run(names[5], k_valu<5>, wps, 64, iters, LDSB, out, stamps);
        run(names[6], k_valu<6>, wps, 64, iters, LDSB, out, stamps);
        run(names[7], k_valu<7>, wps, 64, iters, LDSB, out, stamps);
        run(names[8], k_valu<8>, wps, 64, iters, LDSB, out, stamps);
        run(names[9], k_valu<9>, wps, 64, iters, LDSB, out, stamps);
        run(names[10], k_valu<10>, wps, 64, iters, LDSB, out, stamps);
        run(names[11], k_valu<11>, wps, 128, iters, LDSB, out, stamps);
        run(names[12], k_valu<12>, wps, 64, iters, LDSB, out, stamps);
        run(names[13], k_valu<13>, wps, 64, iters, LDSB, out, stamps);
        run(names[14], k_valu<14>, wps, 64, iters, LDSB, out, stamps);
        run(names[15], k_valu<15>, wps, 64, iters, LDSB, out, stamps);
        run(names[16], k_valu<16>, wps, 64, iters, LDSB, out, stamps);
        std::printf("\n");
    }
    const char *lnames[] = {"ds_read_b64", "ds_read_b128", "ds_write_b64", "ds_write_b32", "ds_read_b32", "ds_bpermute_b32"};
    for (int wps = 1; wps <= 3; ++wps) {
        run(lnames[0], k_lds<0>, wps, 16, iters, LDSB, out, stamps);
        run(lnames[1], k_lds<1>, wps, 16, iters, LDSB, out, stamps);
        run(lnames[2], k_lds<2>, wps, 16, iters, LDSB, out, stamps);
        run(lnames[3], k_lds<3>, wps, 16, iters, LDSB, out, stamps);
        run(lnames[4], k_lds<4>, wps, 16, iters, LDSB, out, stamps);
        run(lnames[5], k_lds<5>, wps, 16, iters, LDSB, out, stamps);
        std::printf("\n");
    }
    return 0;
}
