// Diagnostic phase clock (tools/phase_clock.py): in a build with -DAP_PHASE_CLOCK every wave sums the
// shader cycles (s_memtime) it spends in each phase of a kernel's loop; one table per translation unit.  Never compiled
// into the product library (the macros are empty there).
#pragma once
#if defined(AP_PHASE_CLOCK) && !defined(AP_HOST_EMU)
static __device__ unsigned long long ap_phase_clk[256 * 8 * 12];          // [workgroup][wave][phase]
#define AP_PH_DECL() unsigned long long ap_ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ap_ph_last; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ap_ph_last)::"memory")
#define AP_PH(k) do { unsigned long long ap_ph_t; \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ap_ph_t)::"memory"); \
    ap_ph_acc[k] += ap_ph_t - ap_ph_last; ap_ph_last = ap_ph_t; } while (0)
#define AP_PH_FLUSH() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && threadIdx.x < 512) \
    for (int k = 0; k < 12; ++k) ap_phase_clk[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 12 + k] = ap_ph_acc[k]; } while (0)
#else
#define AP_PH_DECL() do {} while (0)
#define AP_PH(k) do {} while (0)
#define AP_PH_FLUSH() do {} while (0)
#endif

