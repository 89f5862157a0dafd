// TEST INFRASTRUCTURE ONLY.  Builds the product's kernel source for the CPU through
// emu_shim.h and exposes emu_* twins of the C ABI that take HOST pointers.  Same
// validation / geometry code as the HIP build (ap_launch.h), same kernel bodies
// (kernels_generic.h); only the launch mechanism differs.
#include "emu_shim.h"

alignas(16) char ap_smem[160 * 1024];

#include "../../mlx-audio-primitives_amd/csrc/ap_launch.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_generic.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_wave.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_pointwise.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_bigfft.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_ct.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_wave512.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_mel2048.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_frames8.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_stft16.h"
#include "../../mlx-audio-primitives_amd/csrc/kernels_istft16.h"

static thread_local char g_err[512] = "";
char *ap_error_buffer() { return g_err; }
void ap_set_error(const char *msg) { std::snprintf(g_err, sizeof(g_err), "%s", msg); }

template <int EPI, int PADGEN>
static bool emu_launch_ct(ApStftParams &P, int n_fft, int64_t B) {
    int G = 0, lds = 0;
    if (!ap_ct_config(n_fft, EPI == 1 ? P.n_parts : 0, EPI == 1 ? P.n_quads : 0, P.n_mels, &G, &lds)) return false;
    P.tiles_per_clip = (P.T + G - 1) / G;
    int64_t tiles = P.tiles_per_clip * B;
    unsigned grid = (unsigned)(tiles < 3 ? tiles : 3);        // exercise the persistent loop
    if (n_fft == 400) emu_launch(grid, 256, [&] { ap_stft_ct_kernel<EPI, 200, 8, 5, 5, 8, PADGEN, 256>(P); });
    else if (n_fft == 512) emu_launch(grid, 256, [&] { ap_stft_ct_kernel<EPI, 256, 16, 16, 1, 8, PADGEN, 256>(P); });
    else emu_launch(grid, 256, [&] { ap_stft_ct_kernel<EPI, 512, 16, 8, 4, 8, PADGEN, 256>(P); });
    return true;
}

template <int R>
static bool emu_launch_mel8(const ApStftParams &P, int64_t B, const int32_t *plan, const int32_t *desc, float power,
                            unsigned *max_key) {
    ApFrames8Params W;
    int grid = 0;
    const ApFrames8Geom G = {R, ApqGeom<R>::BS, ApqGeom<R>::PS, ApqGeom<R>::WMAX, ApqGeom<R>::WIN_REGS ? 0 : 1};
    if (ap_prepare_frames8(W, P, B, true, plan, desc, APQ_WAVES, G, &grid) != AP_OK) return false;
    if (grid > 1) grid = 1;   // exercise the persistent group loop
    if (max_key) { *max_key = 0x007FFFFFu; W.max_key = max_key; }
    if (!ap_clip_loads_ok(P)) {
        if (power == 2.0f) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_mel8_wave_kernel<R, 2, 1>(W); });
        else if (power == 1.0f) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_mel8_wave_kernel<R, 1, 1>(W); });
        else emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_mel8_wave_kernel<R, 0, 1>(W); });
        return true;
    }
    if (power == 2.0f) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_mel8_wave_kernel<R, 2>(W); });
    else if (power == 1.0f) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_mel8_wave_kernel<R, 1>(W); });
    else emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_mel8_wave_kernel<R, 0>(W); });
    return true;
}

template <int R>
static bool emu_launch_stft8(const ApStftParams &P, int64_t B) {
    ApFrames8Params W;
    int grid = 0;
    const ApFrames8Geom G = {R, ApqGeom<R>::BS, ApqGeom<R>::PS, ApqGeom<R>::WMAX, ApqGeom<R>::WIN_REGS ? 0 : 1};
    if (ap_prepare_frames8(W, P, B, false, nullptr, nullptr, APQ_WAVES, G, &grid) != AP_OK) return false;
    if (grid > 1) grid = 1;
    if (ap_clip_loads_ok(P)) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_stft8_wave_kernel<R, 0>(W); });
    else emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_stft8_wave_kernel<R, 1>(W); });
    return true;
}

struct EmuFftOps {                       // kernel launches of ap_resample_fft_compose on the CPU emulator
    int leg(const ApCfftParams &C, int64_t B) {
        emu_lds_limit(C.tile.lds_bytes), emu_launch((unsigned)(C.tiles_per_signal * B), AP_BLOCK, [&] { ap_cfft_strided_kernel(C); });
        return AP_OK;
    }
    int spectrum(const ap_float2 *X, int64_t Nx, ap_float2 *Y, int64_t num, int64_t B) {
        emu_launch(ap_grid_1d(B * num, AP_BLOCK, kApStreamGrid), AP_BLOCK, [&] { ap_resample_spectrum_kernel(X, Nx, Y, num, B); });
        return AP_OK;
    }
    int chirp_pre(const void *in, int real_in, int64_t N, const float *chirp, int conj, ap_float2 *out, int64_t M, int64_t B) {
        emu_launch(ap_grid_1d(B * M, AP_BLOCK, kApStreamGrid), AP_BLOCK,
                   [&] { ap_chirp_pre_kernel(in, real_in, N, reinterpret_cast<const ap_float2 *>(chirp), conj, out, M, B); });
        return AP_OK;
    }
    int chirp_spec(ap_float2 *buf, const float *spec, int conj, int64_t M, int64_t B) {
        emu_launch(ap_grid_1d(B * M, AP_BLOCK, kApStreamGrid), AP_BLOCK,
                   [&] { ap_chirp_spec_kernel(buf, reinterpret_cast<const ap_float2 *>(spec), conj, M, B); });
        return AP_OK;
    }
    int chirp_post(const ap_float2 *buf, int64_t M, const float *chirp, int conj, int64_t N, float scale, int real_out,
                   void *out, int64_t B) {
        emu_launch(ap_grid_1d(B * N, AP_BLOCK, kApStreamGrid), AP_BLOCK,
                   [&] { ap_chirp_post_kernel(buf, M, reinterpret_cast<const ap_float2 *>(chirp), conj, N, scale, real_out, out, B); });
        return AP_OK;
    }
};

extern "C" {

const char *emu_last_error(void) { return g_err; }

int emu_pad_f32(const float *x, int64_t B, int64_t L, int64_t pad, int mode, float *out) {
    int grid;
    int rc = ap_prepare_pad(x, B, L, pad, mode, out, &grid);
    if (rc != AP_OK) return rc;
    emu_launch(grid, AP_BLOCK, [&] { ap_pad_kernel(x, B, L, pad, mode, out); });
    return AP_OK;
}

int emu_frame_f32(const float *x, int64_t B, int64_t L, int frame_length, int hop, float *out) {
    int grid;
    int64_t T;
    int rc = ap_prepare_frame(x, B, L, frame_length, hop, out, &T, &grid);
    if (rc != AP_OK) return rc;
    emu_launch(grid, AP_BLOCK, [&] { ap_frame_kernel(x, B, L, T, frame_length, hop, out); });
    return AP_OK;
}

int emu_overlap_add_f32(const float *frames, const float *window, int64_t B, int64_t T, int n_fft,
                        int hop, int64_t out_offset, int64_t out_len, float *out) {
    int64_t bpr;
    int rc = ap_prepare_ola(frames, window, B, T, n_fft, hop, out_offset, out_len, out, &bpr);
    if (rc != AP_OK) return rc;
    emu_launch((unsigned)(bpr * B), AP_BLOCK, [&] {
        ap_overlap_add_kernel(frames, window, T, n_fft, hop, out_offset, out_len, bpr, out);
    });
    return AP_OK;
}

int emu_stft_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                 const float *tw, int center, int pad_mode, int64_t T, float *out) {
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, n_fft, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    P.out_c = reinterpret_cast<ap_float2 *>(out);
    if (n_fft == 2048) {
        ApStftWaveParams W;
        int grid = 0;
        if (ap_prepare_stft_wave(W, P, B, &grid) == AP_OK) {
            if (grid > 2) grid = 2;   // exercise the persistent group loop
            if (ap_clip_loads_ok(W))
                emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_wave_kernel<0>(W); });
            else
                emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_wave_kernel<1>(W); });
            return AP_OK;
        }
    }
    if (n_fft == 1024) {
        ApStftWave512Params W;
        int grid = 0;
        if (ap_prepare_stft_wave512(W, P, B, APHS_WAVES, APH_X_COMPLEX, APHS_OB_ROWS * APHS_OB_ROW, &grid) == AP_OK) {
            if (grid > 2) grid = 2;   // exercise the persistent group loop and the carries
            emu_lds_limit(W.lds_bytes), (W.padgen ? emu_launch((unsigned)grid, 64 * APHS_WAVES, [&] { ap_stft1024_wave_kernel<1>(W); }) : emu_launch((unsigned)grid, 64 * APHS_WAVES, [&] { ap_stft1024_wave_kernel<0>(W); }));
            return AP_OK;
        }
    }
    if (n_fft == 400 || n_fft == 512 || n_fft == 256) {
        if (n_fft == 400 ? emu_launch_stft8<25>(P, B) : n_fft == 512 ? emu_launch_stft8<32>(P, B) : emu_launch_stft8<16>(P, B))
            return AP_OK;
    }
    if (ap_clip_loads_ok(P) ? emu_launch_ct<0, 0>(P, n_fft, B) : emu_launch_ct<0, 1>(P, n_fft, B))
        return AP_OK;
    emu_lds_limit(P.tile.lds_bytes), emu_launch((unsigned)(P.tiles_per_clip * B), AP_BLOCK, [&] { ap_stft_generic_kernel<0>(P); });
    return AP_OK;
}

// n_fft = 2048 STFT on the 16-frames-per-group kernel (kernels_stft16.h); `out` has rows Ts complex apart
int emu_stft16_f32(const float *y, int64_t B, int64_t L, int hop, const float *window, const float *tw,
                   int center, int pad_mode, int64_t T, int64_t Ts, float *out, int grid_cap, int force_unaligned) {
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, 2048, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    P.out_c = reinterpret_cast<ap_float2 *>(out);
    ApStft16Params W;
    int grid = 0, aligned = 0;
    if (ap_prepare_stft16(W, P, B, Ts, &grid, &aligned) != AP_OK) return AP_ERR_UNSUPPORTED;
    if (force_unaligned == 1) aligned = 0;
    if (grid > grid_cap) grid = grid_cap;   // exercise the persistent group loop and the carries
    const bool pg = !ap_clip_loads_ok(W);
    emu_lds_limit(W.lds_bytes);
    // force_unaligned: 1 = the carry path on an aligned layout; 3 = the whole-group tile (T2 = 1)
    if (aligned && force_unaligned == 3) {
        if (pg) emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<1, 1, 0, 0, 1>(W); });
        else emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<0, 1, 0, 0, 1>(W); });
    } else if (aligned) {
        if (pg) emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<1, 1, 0>(W); });
        else emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<0, 1, 0>(W); });
    } else {
        if (pg) emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<1, 0, 0>(W); });
        else emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<0, 0, 0>(W); });
    }
    return aligned;
}

// The STFT with the Griffin-Lim projection in its store phase (kernels_stft16.h, GL = 1: 8-byte accesses, GL = 2: two
// frames per thread as 16-byte accesses): raw spectrum -> out, rebuilt = S unit(raw) + m (S unit(raw) - S unit(prev)),
// all complex arrays with rows Ts apart, `mag` dense (B, 1025, T)
int emu_stft16_gl_f32(const float *y, int64_t B, int64_t L, int hop, const float *window, const float *tw,
                      int center, int pad_mode, int64_t T, int64_t Ts, const float *prev, const float *mag, float momentum,
                      float *out, float *rebuilt, int grid_cap, int variant) {
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, 2048, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    P.out_c = reinterpret_cast<ap_float2 *>(out);
    ApStft16Params W;
    int grid = 0, aligned = 0;
    if (ap_prepare_stft16(W, P, B, Ts, &grid, &aligned) != AP_OK || !aligned) return AP_ERR_UNSUPPORTED;
    W.gl_prev = reinterpret_cast<const ap_float2 *>(prev);
    W.gl_mag = mag;
    W.gl_rebuilt = reinterpret_cast<ap_float2 *>(rebuilt);
    W.gl_momentum = momentum;
    if (grid > grid_cap) grid = grid_cap;
    const bool pg = !ap_clip_loads_ok(W);
    emu_lds_limit(W.lds_bytes);
    if (variant == 2) {
        if (pg) emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<1, 1, 0, 2>(W); });
        else emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<0, 1, 0, 2>(W); });
    } else {
        if (pg) emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<1, 1, 0, 1>(W); });
        else emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_stft2048_g16_kernel<0, 1, 0, 1>(W); });
    }
    return AP_OK;
}

int emu_melspec_f32(const float *y, int64_t B, int64_t L, int n_fft, int hop, const float *window,
                    const float *tw, int center, int pad_mode, int64_t T, const float *fb,
                    const int32_t *plan, const int32_t *desc, int n_mels, float power, float *out,
                    unsigned *max_key) {
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, n_fft, hop, window, tw, center, pad_mode, T);
    if (rc != AP_OK) return rc;
    rc = ap_prepare_mel(P, fb, plan, desc, n_mels, power, out);
    if (rc != AP_OK) return rc;
    if (ap_mel_wave_eligible(n_fft, plan, desc)) {
        ApMelWaveParams W;
        int grid = 0, n_pass = 0;
        // the run kernel where it applies (desc[0] & 512: test-only flag that keeps the tile kernel)
        if (!(desc[0] & 512) && (power == 2.0f || power == 1.0f) &&
            ap_prepare_mel_run(W, P, B, plan, desc, APM_WAVES, APW_X_COMPLEX, APM_PARTIAL_OFF, &n_pass, &grid) == AP_OK) {
            if (grid > 1) grid = 1;   // exercise the persistent frame loop
            if (max_key) { *max_key = 0x007FFFFFu; W.max_key = max_key; }
            const bool padgen = !ap_clip_loads_ok(P);
#define EMU_RUN(PM, NP) do { if (W.hopj == 4 && !padgen) emu_launch((unsigned)grid, 64 * APM_WAVES, [&] { ap_mel2048_run_kernel<PM, NP, 4>(W); }); \
                             else if (W.hopj == 4) emu_launch((unsigned)grid, 64 * APM_WAVES, [&] { ap_mel2048_run_kernel<PM, NP, 4, 2>(W); }); \
                             else if (!padgen) emu_launch((unsigned)grid, 64 * APM_WAVES, [&] { ap_mel2048_run_kernel<PM, NP, 0>(W); }); \
                             else emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APM_WAVES, [&] { ap_mel2048_run_kernel<PM, NP, 0, 2>(W); }); } while (0)
            emu_lds_limit(W.lds_bytes);
            if (power == 2.0f) { if (n_pass == 1) EMU_RUN(2, 1); else if (n_pass == 2) EMU_RUN(2, 2); else if (n_pass == 3) EMU_RUN(2, 3); else EMU_RUN(2, 4); }
            else { if (n_pass == 1) EMU_RUN(1, 1); else if (n_pass == 2) EMU_RUN(1, 2); else if (n_pass == 3) EMU_RUN(1, 3); else EMU_RUN(1, 4); }
#undef EMU_RUN
            return AP_OK;
        }
        if (ap_prepare_mel_wave(W, P, B, plan, desc, &grid) == AP_OK) {
            if (grid > 1) grid = 1;   // exercise the persistent tile loop
            if (max_key) { *max_key = 0x007FFFFFu; W.max_key = max_key; }
            const bool gen = !ap_clip_loads_ok(W);
            if (power == 2.0f && !gen) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APW_WAVES, [&] { ap_mel2048_wave_kernel<2, 0>(W); });
            else if (power == 2.0f) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APW_WAVES, [&] { ap_mel2048_wave_kernel<2, 1>(W); });
            else if (power == 1.0f && !gen) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APW_WAVES, [&] { ap_mel2048_wave_kernel<1, 0>(W); });
            else if (power == 1.0f) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APW_WAVES, [&] { ap_mel2048_wave_kernel<1, 1>(W); });
            else if (!gen) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APW_WAVES, [&] { ap_mel2048_wave_kernel<0, 0>(W); });
            else emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APW_WAVES, [&] { ap_mel2048_wave_kernel<0, 1>(W); });
            return AP_OK;
        }
    }
    if ((n_fft == 400 || n_fft == 512 || n_fft == 256) && !(desc && (desc[0] & 512))) {
        if (n_fft == 400 ? emu_launch_mel8<25>(P, B, plan, desc, power, max_key)
            : n_fft == 512 ? emu_launch_mel8<32>(P, B, plan, desc, power, max_key)
                           : emu_launch_mel8<16>(P, B, plan, desc, power, max_key))
            return AP_OK;
    }
    if (n_fft == 1024) {
        ApMelWave512Params W;
        int grid = 0;
        if (ap_prepare_mel_wave512(W, P, B, plan, desc, APH_WAVES, APH_X_COMPLEX, APH_PASSES, &grid) == AP_OK) {
            if (grid > 1) grid = 1;   // exercise the persistent frame loop
            if (max_key) { *max_key = 0x007FFFFFu; W.max_key = max_key; }
            emu_lds_limit(W.lds_bytes);
            const bool padgen = !ap_clip_loads_ok(P);
#define EMU_M1024(PM) do { if (hop == 256 && !padgen) emu_launch((unsigned)grid, 64 * APH_WAVES, [&] { ap_mel1024_wave_kernel<PM, 2, 0>(W); }); \
                           else if (hop == 256) emu_launch((unsigned)grid, 64 * APH_WAVES, [&] { ap_mel1024_wave_kernel<PM, 2, 1>(W); }); \
                           else if (!padgen) emu_launch((unsigned)grid, 64 * APH_WAVES, [&] { ap_mel1024_wave_kernel<PM, 0, 0>(W); }); \
                           else emu_launch((unsigned)grid, 64 * APH_WAVES, [&] { ap_mel1024_wave_kernel<PM, 0, 1>(W); }); } while (0)
            if (power == 2.0f) EMU_M1024(2); else if (power == 1.0f) EMU_M1024(1); else EMU_M1024(0);
#undef EMU_M1024
            return AP_OK;
        }
    }
    if (!(desc && (desc[0] & AP_PLAN_FORCE_GENERIC))) {
        if (ap_clip_loads_ok(P) ? emu_launch_ct<1, 0>(P, n_fft, B) : emu_launch_ct<1, 1>(P, n_fft, B))
            return AP_OK;
    }
    emu_lds_limit(P.tile.lds_bytes), emu_launch((unsigned)(P.tiles_per_clip * B), AP_BLOCK, [&] { ap_stft_generic_kernel<1>(P); });
    return AP_OK;
}

// geometry the launch code derives for the n_fft = 2048 mel wave kernel: [0] partial_stride,
// [1] lds_bytes, [2] n_slots, [3] grid; returns 1 when the kernel does not apply
int emu_mel_wave_geometry(const float *fb, const int32_t *plan, const int32_t *desc, int n_mels,
                          int64_t B, int64_t L, int hop, int32_t *out4) {
    static float dummy[4];
    ApStftParams P;
    const int64_t T = ap_n_frames(L, 2048, hop, 1);
    int rc = ap_prepare_stft(P, dummy, B, L, 2048, hop, dummy, dummy, 1, AP_PAD_CONSTANT, T);
    if (rc != AP_OK) return rc;
    rc = ap_prepare_mel(P, fb, plan, desc, n_mels, 2.0f, dummy);
    if (rc != AP_OK) return rc;
    if (!ap_mel_wave_eligible(2048, plan, desc)) return 1;
    ApMelWaveParams W;
    int grid = 0;
    if (ap_prepare_mel_wave(W, P, B, plan, desc, &grid) != AP_OK) return 1;
    out4[0] = W.partial_stride; out4[1] = W.lds_bytes; out4[2] = W.n_slots; out4[3] = grid;
    return AP_OK;
}

int emu_irfft_frames_f32(const float *S, int64_t B, int64_t T, int n_fft, const float *tw,
                         float *frames) {
    ApIrfftParams P;
    int rc = ap_prepare_irfft(P, S, B, T, n_fft, tw, frames);
    if (rc != AP_OK) return rc;
    if (n_fft == 2048) {
        ApIrfftWaveParams W;
        int grid = 0;
        if (ap_prepare_irfft_wave(W, P, B, &grid) == AP_OK) {
            if (grid > 2) grid = 2;
            emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_irfft2048_wave_kernel<0>(W); });
            return AP_OK;
        }
    }
    if (n_fft == 512 || n_fft == 400 || n_fft == 256) {
        ApIrfft8Params W;
        int grid = 0;
        const int R = n_fft / 16;
        const int bs = R == 25 ? ApqGeom<25>::BS : R == 32 ? ApqGeom<32>::BS : ApqGeom<16>::BS;
        if (ap_prepare_irfft8(W, P, B, R, bs, APQ_WAVES, &grid) == AP_OK) {
            if (grid > 1) grid = 1;
            if (R == 25) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_irfft8_wave_kernel<25>(W); });
            else if (R == 32) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_irfft8_wave_kernel<32>(W); });
            else emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_irfft8_wave_kernel<16>(W); });
            return AP_OK;
        }
    }
    emu_lds_limit(P.tile.lds_bytes), emu_launch((unsigned)(P.tiles_per_clip * B), AP_BLOCK, [&] { ap_irfft_generic_kernel(P); });
    return AP_OK;
}

static int emu_decim2 = 1;         // 0: keep the four-outputs-per-thread decimator (tests run both)
void emu_set_decim2(int on) { emu_decim2 = on; }

int emu_resample_poly_f32(const float *x, int64_t B, int64_t L, int up, int down, const float *taps,
                          int n_taps, int n_pre_remove, int64_t n_out, float *out) {
    int64_t bpr;
    int rc = ap_prepare_resample_poly(x, B, L, up, down, taps, n_taps, n_pre_remove, n_out, out, &bpr);
    if (rc != AP_OK) return rc;
    int lds = 0, Q = 0;
    if (emu_decim2 && ap_resample_decim2_eligible(up, down, n_taps, &Q, &lds)) {
        const int64_t per_block = AP_BLOCK * 2 * Q;
        const int64_t bprq = (n_out + per_block - 1) / per_block;
        emu_lds_limit(lds), emu_launch((unsigned)(bprq * B), AP_BLOCK, [&] {
            ap_resample_decim2_kernel<4>(x, L, down, taps, n_taps, n_pre_remove, n_out, bprq, out);
        });
        return AP_OK;
    }
    if (ap_resample_decim_eligible(up, down, n_taps, &Q, &lds)) {
        const int64_t per_block = AP_BLOCK * 4 * Q;
        const int64_t bprq = (n_out + per_block - 1) / per_block;
        emu_launch((unsigned)(bprq * B), AP_BLOCK, [&] {
            if (Q == 4) ap_resample_decim_kernel<4>(x, L, down, taps, n_taps, n_pre_remove, n_out, bprq, out);
            else if (Q == 2) ap_resample_decim_kernel<2>(x, L, down, taps, n_taps, n_pre_remove, n_out, bprq, out);
            else ap_resample_decim_kernel<1>(x, L, down, taps, n_taps, n_pre_remove, n_out, bprq, out);
        });
        return AP_OK;
    }
    if (up > 1) {
        const int K = (n_taps + up - 1) / up, KS = K | 1;
        const int64_t per_block = (int64_t)AP_BLOCK * AP_RSPL_R;
        const int64_t span = (per_block * down) / up + K + 4;
        const int64_t lds = ((int64_t)up * KS + span) * (int64_t)sizeof(float);
        const int64_t bprl = (n_out + per_block - 1) / per_block;
        if (lds <= 64 * 1024) {
            emu_launch((unsigned)(bprl * B), AP_BLOCK, [&] {
                ap_resample_poly_lds_kernel(x, L, up, down, taps, n_taps, n_pre_remove, n_out, bprl, K, KS, (int)span, out);
            });
            return AP_OK;
        }
    }
    emu_launch((unsigned)(bpr * B), AP_BLOCK, [&] {
        ap_resample_poly_kernel(x, L, up, down, taps, n_taps, n_pre_remove, n_out, bpr, out);
    });
    return AP_OK;
}

int emu_resample_linear_f32(const float *x, int64_t B, int64_t L, int64_t n_out, double scale, float *out) {
    emu_launch(ap_grid_1d(B * n_out, AP_BLOCK, kApStreamGrid), AP_BLOCK, [&] {
        ap_resample_linear_kernel(x, B, L, n_out, scale, out);
    });
    return AP_OK;
}

int emu_gl_project_f32(int mode, const float *S, const float *angles, const float *R, int64_t TR,
                       int64_t BF, int64_t T, float momentum, float *tprev, float *rebuilt) {
    emu_launch(ap_grid_1d(BF * T, AP_BLOCK, kApStreamGrid), AP_BLOCK, [&] {
        ap_gl_project_kernel(mode, S, angles, reinterpret_cast<const ap_float2 *>(R), TR, BF, T, momentum,
                             reinterpret_cast<ap_float2 *>(tprev), reinterpret_cast<ap_float2 *>(rebuilt));
    });
    return AP_OK;
}

// fused irfft + overlap-add (n_fft = 2048); grid_cap > 0 limits the workgroups so stretches get long
int emu_istft1024_fused_f32(const float *S, int64_t B, int64_t T, int hop, const float *window, const float *tw,
                            int64_t out_offset, int64_t out_len, int grid_cap, float *out) {
    ApIstftWave512Params W;
    int grid = 0;
    if (ap_prepare_istft_wave512(W, S, tw, B, T, window, hop, out_offset, out_len, out, APHS_WAVES, APH_X_COMPLEX,
                                 APHS_OB_ROWS * APHS_OB_ROW, &grid) != AP_OK) {
        // below the product's size threshold: run the fused kernel anyway for coverage
        if ((hop != 128 && hop != 256 && hop != 512) || out_offset % 4 != 0) return AP_ERR_UNSUPPORTED;
        W.S = reinterpret_cast<const ap_float2 *>(S); W.tw = reinterpret_cast<const ap_float2 *>(tw);
        W.window = window; W.y = out; W.T = T;
        W.groups_per_clip = (T + 7) / 8; W.n_groups = W.groups_per_clip * B;
        W.out_offset = out_offset; W.out_len = out_len; W.hop = hop;
        int off = APHS_WAVES * APH_X_COMPLEX * 8;
        W.off_tw1 = off; off += 8 * 64 * 8;
        W.off_tw2 = off; off += 64 * 8;
        W.off_win = off; off += 1024 * 4;
        W.off_ib = off; off += ap_align16(APHS_OB_ROWS * APHS_OB_ROW * 8);
        W.off_carry = off; off += 2 * (1024 - hop) * 4;
        W.lds_bytes = off;
        grid = (int)(W.n_groups < 256 ? W.n_groups : 256);
    }
    if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
    emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APHS_WAVES, [&] { ap_istft1024_wave_kernel(W); });
    return AP_OK;
}

// fused ISTFT of the frames8 family (n_fft 512 / 400 / 256); grid_cap > 0 limits the workgroups
int emu_istft8_fused_f32(const float *S, int64_t B, int64_t T, int n_fft, int hop, const float *window, const float *tw,
                         int64_t out_offset, int64_t out_len, int grid_cap, float *out) {
    ApIstft8Params W;
    int grid = 0;
    if (ap_prepare_istft8(W, S, tw, B, T, n_fft, window, hop, out_offset, out_len, out, APQ_WAVES, &grid) != AP_OK)
        return AP_ERR_UNSUPPORTED;
    if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
    if (n_fft == 512) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_istft8_wave_kernel<32>(W); });
    else if (n_fft == 400) emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_istft8_wave_kernel<25>(W); });
    else emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APQ_WAVES, [&] { ap_istft8_wave_kernel<16>(W); });
    return AP_OK;
}

int emu_istft_fused_f32(const float *S, int64_t B, int64_t T, int hop, const float *window, const float *tw,
                        int64_t out_offset, int64_t out_len, int grid_cap, float *out) {
    ApIrfftParams P;
    int rc = ap_prepare_irfft(P, S, B, T, 2048, tw, out /* unused frames pointer, non-NULL */);
    if (rc != AP_OK) return rc;
    ApIrfftWaveParams W;
    int grid = 0;
    if (ap_prepare_istft_wave(W, P, B, window, hop, out_offset, out_len, out, &grid) != AP_OK) {
        // below the product's size threshold: still run the fused kernel for coverage
        int g0 = 0;
        if (ap_prepare_irfft_wave(W, P, B, &g0) != AP_OK) return AP_ERR_UNSUPPORTED;
        if (hop < 256 || 2048 % hop != 0 || out_offset % 4 != 0) return AP_ERR_UNSUPPORTED;
        W.window = window; W.y = out; W.hop = hop; W.out_offset = out_offset; W.out_len = out_len;
        int off = W.lds_bytes;
        W.off_win = off; off += 2048 * 4;
        W.off_carry = off; off += 2 * (2048 - hop) * 4;
        W.lds_bytes = off;
        grid = (int)(W.n_groups < 256 ? W.n_groups : 256);
    }
    if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
    emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_irfft2048_wave_kernel<1>(W); });
    return AP_OK;
}

// n_fft = 2048 fused ISTFT with 16-frame loads (kernels_istft16.h); S has rows Ts complex apart
int emu_istft16_f32(const float *S, int64_t B, int64_t T, int64_t Ts, int hop, const float *window, const float *tw,
                    int64_t out_offset, int64_t out_len, int grid_cap, int variant, float *out) {
    ApIstft16Params W;
    int grid = 0;
    if (ap_prepare_istft16(W, S, tw, B, T, Ts, window, hop, out_offset, out_len, out, &grid) != AP_OK)
        return AP_ERR_UNSUPPORTED;
    const int64_t n_steps = ((T + 7) / 8) * B;                // the kernel's stretches are in 8-frame steps
    if (grid_cap > 0) grid = grid_cap < n_steps ? grid_cap : (int)n_steps;
    emu_lds_limit(W.lds_bytes);
    // variant 0: the product's default (two-round staging pass, the next group's loads issued inside the first step's
    // transform); 1: eight rounds; 2 / 3: the same two with the loads issued in the staging pass
#define EMU_ISTFT16(SP, TL) do { \
    if (hop == 256) emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_istft2048_g16_kernel<8, true, SP, TL>(W); }); \
    else if (hop == 512) emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_istft2048_g16_kernel<9, true, SP, TL>(W); }); \
    else emu_launch((unsigned)grid, 64 * APS_WAVES, [&] { ap_istft2048_g16_kernel<10, true, SP, TL>(W); }); } while (0)
    if (variant == 0) EMU_ISTFT16(1, 1);
    else if (variant == 1) EMU_ISTFT16(1, 0);
    else if (variant == 2) EMU_ISTFT16(0, 1);
    else EMU_ISTFT16(0, 0);
#undef EMU_ISTFT16
    return AP_OK;
}

int emu_to_db_f32(const float *S, int64_t n, float coef, float amin, float ref_value, int ref_is_max,
                  float top_db, float *out) {
    unsigned keys[2] = {0x007FFFFFu, 0x007FFFFFu};
    const int grid = ap_grid_1d(n, AP_BLOCK, kApStreamGrid);
    if (ref_is_max) emu_launch(grid, AP_BLOCK, [&] { ap_reduce_max_kernel(S, n, &keys[1]); });
    const bool clip = top_db >= 0.0f;
    if (clip) emu_launch(grid, AP_BLOCK, [&] { ap_reduce_max_kernel(S, n, &keys[0]); });
    ApDbParams D;
    D.coef = coef; D.amin = amin; D.ref_value = ref_value; D.top_db = clip ? top_db : -1.0f;
    D.ref_key = ref_is_max ? &keys[1] : nullptr; D.smax_key = &keys[0];
    emu_launch(grid, AP_BLOCK, [&] { ap_to_db_kernel(S, n, D, out); });
    return AP_OK;
}

// db = 1: fused power_to_db (ref_value, amin, top_db) + DCT, like ap_db_dct_f32
static int emu_dct_wide = 0;      // 1: the 64-bit-offset instantiation of ap_dct_kernel
void emu_set_dct_wide(int on) { emu_dct_wide = on; }

int emu_dct_f32(const float *x, const float *C, const float *row_scale, int64_t outer, int n_in,
                int64_t inner, int n_out, int db, float coef, float amin, float ref_value, float top_db,
                float *out) {
    const int grid = ap_grid_1d(outer * inner, AP_BLOCK, kApStreamGrid);
    unsigned key = 0x007FFFFFu;
    ApDbParams D = {};
    if (db) {
        const bool clip = top_db >= 0.0f;
        if (clip) emu_launch(grid, AP_BLOCK, [&] { ap_reduce_max_kernel(x, outer * n_in * inner, &key); });
        D.coef = coef; D.amin = amin; D.ref_value = ref_value; D.top_db = clip ? top_db : -1.0f;
        D.ref_key = nullptr; D.smax_key = &key;
    }
    const int KT = n_out <= 16 ? 16 : 32;
    if (n_in * KT * 4 > 64 * 1024) {
        if (db) return AP_ERR_UNSUPPORTED;
        if (KT == 16) emu_launch(grid, AP_BLOCK, [&] { ap_dct_generic_kernel<16>(x, C, row_scale, outer, n_in, inner, n_out, out); });
        else emu_launch(grid, AP_BLOCK, [&] { ap_dct_generic_kernel<32>(x, C, row_scale, outer, n_in, inner, n_out, out); });
        return AP_OK;
    }
#define EMU_DCT(KTV, DBV) \
    do { \
        if (emu_dct_wide) emu_launch(grid, AP_BLOCK, [&] { ap_dct_kernel<KTV, DBV, 1>(x, C, row_scale, outer, n_in, inner, n_out, D, out); }); \
        else emu_launch(grid, AP_BLOCK, [&] { ap_dct_kernel<KTV, DBV, 0>(x, C, row_scale, outer, n_in, inner, n_out, D, out); }); \
    } while (0)
    if (KT == 16 && db) EMU_DCT(16, 1);
    else if (KT == 16) EMU_DCT(16, 0);
    else if (db) EMU_DCT(32, 1);
    else EMU_DCT(32, 0);
#undef EMU_DCT
    return AP_OK;
}

int emu_resample_fft_chirp_f32(const float *x, int64_t B, int64_t Nx, int64_t num, int64_t Mx, const float *tw_x1,
                               const float *tw_x2, const float *chirp_x, const float *spec_x, int64_t My,
                               const float *tw_y1, const float *tw_y2, const float *chirp_y, const float *spec_y,
                               float *ws, float *out) {
    const ApCfftSide X = {Nx, Mx, tw_x1, tw_x2, chirp_x, spec_x};
    const ApCfftSide Y = {num, My, tw_y1, tw_y2, chirp_y, spec_y};
    EmuFftOps ops;
    return ap_resample_fft_compose(ops, x, B, X, Y, ws, out);
}

int emu_resample_fft_f32(const float *x, int64_t B, int64_t Nx, int64_t num, const float *tw_x1,
                         const float *tw_x2, const float *tw_y1, const float *tw_y2, float *ws, float *out) {
    return emu_resample_fft_chirp_f32(x, B, Nx, num, 0, tw_x1, tw_x2, nullptr, nullptr, 0, tw_y1, tw_y2, nullptr,
                                      nullptr, ws, out);
}

int emu_spectral_audio_f32(const float *y, int64_t B, int64_t L, int hop, const float *window, const float *tw,
                           int center, int64_t T, const float *freq, float power, float p, int norm,
                           float roll_percent, float amin, float *centroid, float *bandwidth, float *rolloff,
                           float *flatness) {
    ApStftParams P;
    int rc = ap_prepare_stft(P, y, B, L, 2048, hop, window, tw, center, AP_PAD_CONSTANT, T);
    if (rc != AP_OK) return rc;
    ApSpecWaveParams W;
    int grid = 0;
    if (ap_prepare_spec_run(W, P, B, APM_WAVES, APW_X_COMPLEX, &grid) != AP_OK)
        AP_FAIL(AP_ERR_UNSUPPORTED, "spectral features from audio: shape not served");
    if (grid > 1) grid = 1;
    W.freq = freq;
    W.centroid = centroid; W.bandwidth = bandwidth; W.rolloff = rolloff; W.flatness = flatness;
    W.power = power; W.p = p; W.norm = norm; W.roll_percent = roll_percent; W.amin = amin;
#define EMU_SPEC(PM, FL) do { if (W.hopj == 4 && p == 2.0f) emu_launch((unsigned)grid, 64 * APM_WAVES, [&] { ap_spec2048_run_kernel<PM, 4, FL, 0>(W); }); \
                              else if (W.hopj == 4) emu_launch((unsigned)grid, 64 * APM_WAVES, [&] { ap_spec2048_run_kernel<PM, 4, FL, 1>(W); }); \
                              else if (p == 2.0f) emu_launch((unsigned)grid, 64 * APM_WAVES, [&] { ap_spec2048_run_kernel<PM, 0, FL, 0>(W); }); \
                              else emu_lds_limit(W.lds_bytes), emu_launch((unsigned)grid, 64 * APM_WAVES, [&] { ap_spec2048_run_kernel<PM, 0, FL, 1>(W); }); } while (0)
    emu_lds_limit(W.lds_bytes);
    if (flatness) { if (power == 2.0f) EMU_SPEC(2, 1); else if (power == 1.0f) EMU_SPEC(1, 1); else EMU_SPEC(0, 1); }
    else { if (power == 2.0f) EMU_SPEC(2, 0); else if (power == 1.0f) EMU_SPEC(1, 0); else EMU_SPEC(0, 0); }
#undef EMU_SPEC
    return AP_OK;
}

int emu_cfft_split(int64_t N, int *N1, int *N2) { return ap_cfft_split(N, N1, N2); }

// launches that wrote past the dynamic LDS their host code asked for (emu_shim.h), since the library was loaded
int emu_lds_overrun_count() { return emu_lds_overruns; }

// proves the guard sees an overrun: a launch that was promised 1024 bytes of LDS writes byte 2000; returns the
// number of overruns it added (1) and takes it off the count again
int emu_lds_guard_selftest() {
    const int before = emu_lds_overruns;
    emu_lds_limit(1024);
    emu_launch(1, 64, [&] { if (threadIdx.x == 0) ap_smem[2000] = 1; });
    const int seen = emu_lds_overruns - before;
    emu_lds_overruns = before;
    return seen;
}

int emu_pcg64_uniform_f32(unsigned long long st_hi, unsigned long long st_lo, unsigned long long inc_hi,
                          unsigned long long inc_lo, double low, double high, int64_t n, float *out) {
    emu_launch(ap_grid_1d(n, AP_BLOCK, kApStreamGrid), AP_BLOCK, [&] {
        ap_pcg64_uniform_kernel(st_hi, st_lo, inc_hi, inc_lo, low, high - low, n, out);
    });
    return AP_OK;
}

int emu_complex_unary_f32(const float *S, int64_t n, int mode, float *out) {
    emu_launch(ap_grid_1d(n, AP_BLOCK, kApStreamGrid), AP_BLOCK, [&] {
        ap_complex_unary_kernel(reinterpret_cast<const ap_float2 *>(S), n, mode, out);
    });
    return AP_OK;
}

}  // extern "C"
