// Launchers that live in translation units of their own (the library is compiled as several
// objects in parallel, _build.py); audioprims.hip validates and calls them.
#pragma once
#include "ap_launch.h"

// stft16.hip: n_fft = 2048 STFT on the 16-frames-per-group kernel (kernels_stft16.h).
// Returns AP_OK, an error status, or 1 when the shape is not served (the caller falls back).
int ap_launch_stft16(const ApStftParams &P, int64_t B, int64_t Ts, void *stream);

int ap_launch_stft16_gl(const ApStftParams &P, int64_t B, int64_t Ts, const float *prev, const float *mag, float momentum,
                        float *rebuilt, void *stream);

// istft16.hip: fused n_fft = 2048 ISTFT with 16-frame loads (kernels_istft16.h); S has rows Ts complex apart.
// Returns AP_OK, an error status, or 1 when the shape is not served.
int ap_launch_istft16(const float *S, const float *tw, int64_t B, int64_t T, int64_t Ts, const float *window, int hop,
                      int64_t out_offset, int64_t out_len, float *out, void *stream);

// Upper bound on the persistent grids of the two launchers above (0 = none): the Griffin-Lim loop runs chains on
// several streams and may want their kernels side by side on disjoint CUs instead of queued behind each other.
extern thread_local int ap_g16_grid_cap;
