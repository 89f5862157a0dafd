// TEST INFRASTRUCTURE ONLY — a minimal SIMT emulator so the *same kernel source*
// (mlx-audio-primitives_amd/csrc/kernels_*.h) can be executed on the CPU: one OS
// thread per GPU thread of a workgroup, std::barrier for __syncthreads(),
// workgroups run one after another.  Used by tests/test_emu_kernels.py (and for
// ASan/UBSan runs, which the GPU pool does not allow).  Never loaded by the
// product package.
#pragma once
#include <barrier>
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

struct emu_dim3 {
    unsigned x = 1, y = 1, z = 1;
};

inline thread_local emu_dim3 threadIdx, blockIdx;
inline emu_dim3 blockDim, gridDim;
inline std::barrier<> *emu_barrier_ptr = nullptr;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
#define __launch_bounds__(...)
#define AP_DEV inline
#define AP_HOST_EMU 1

// ---- wave-level helpers used by kernels_wave.h: wave w = threads [64w, 64w+64) -----
inline float emu_xchg[1024];
inline std::barrier<> *emu_wave_barriers[16];
inline void __syncthreads() { emu_barrier_ptr->arrive_and_wait(); }
inline void emu_wave_sync() { emu_wave_barriers[threadIdx.x >> 6]->arrive_and_wait(); }
inline float emu_lane_xor(float x, int mask) {
    emu_xchg[threadIdx.x] = x;
    emu_wave_sync();
    const float y = emu_xchg[threadIdx.x ^ mask];
    emu_wave_sync();
    return y;
}
inline float emu_lane_perm(float x, int src_lane) {      // value of lane `src_lane` of the same wave
    emu_xchg[threadIdx.x] = x;
    emu_wave_sync();
    const float y = emu_xchg[(threadIdx.x & ~63u) | (unsigned)(src_lane & 63)];
    emu_wave_sync();
    return y;
}
inline float ap_quad_xor1(float x) { return emu_lane_xor(x, 1); }
inline float ap_quad_xor2(float x) { return emu_lane_xor(x, 2); }

alignas(16) inline char ap_smem_storage[160 * 1024];
// kernels declare:  extern __shared__ __attribute__((aligned(16))) char ap_smem[];
// -> after the macros above that is an extern char array; define it here.
extern char ap_smem[];

// Dynamic-LDS guard: a launcher that knows the kernel's lds_bytes calls emu_lds_limit(bytes) first; everything
// past that offset is poisoned for the launch and checked afterwards (a kernel writing beyond the LDS the host
// asked for corrupts a neighbour or faults on the GPU; here it would go unnoticed inside the 160 KB array).
inline int emu_lds_limit_bytes = 0;
inline int emu_lds_overruns = 0;
inline void emu_lds_limit(int bytes) { emu_lds_limit_bytes = bytes; }

template <class F>
void emu_launch(unsigned grid, unsigned block, F &&body) {
    const int guard_from = emu_lds_limit_bytes;
    emu_lds_limit_bytes = 0;
    if (guard_from > 0)
        for (int i = guard_from; i < 160 * 1024; ++i) ap_smem[i] = (char)0xA5;
    gridDim.x = grid;
    blockDim.x = block;
    std::barrier<> bar((std::ptrdiff_t)block);
    emu_barrier_ptr = &bar;
    std::vector<std::unique_ptr<std::barrier<>>> wbars;
    for (unsigned w = 0; w * 64 < block; ++w) {
        const unsigned n = block - w * 64 < 64 ? block - w * 64 : 64;
        wbars.emplace_back(new std::barrier<>((std::ptrdiff_t)n));
        emu_wave_barriers[w] = wbars.back().get();
    }
    std::vector<std::thread> threads;
    threads.reserve(block);
    for (unsigned t = 0; t < block; ++t) {
        threads.emplace_back([&, t]() {
            threadIdx.x = t;
            for (unsigned b = 0; b < grid; ++b) {
                blockIdx.x = b;
                body();
                bar.arrive_and_wait();   // workgroups of one launch share ap_smem
            }
        });
    }
    for (auto &th : threads) th.join();
    emu_barrier_ptr = nullptr;
    if (guard_from > 0)
        for (int i = guard_from; i < 160 * 1024; ++i)
            if (ap_smem[i] != (char)0xA5) { ++emu_lds_overruns; break; }
}
