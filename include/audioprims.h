/*
 * audioprims.h — C ABI of libaudioprims_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary that takes the place of the reference's native
 * extension (nanobind module `_ext`, /root/reference/csrc/bindings.cpp:11-485)
 * and of the two third-party device calls on the hot path that have no source
 * under the reference (`mx.fft.rfft`, stft.py:130; `mx.fft.irfft`, stft.py:295).
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures.  `stream` is a hipStream_t
 *    passed as void* (NULL = the default stream).
 *  - every pointer marked [dev] is device memory owned and pre-allocated by the
 *    caller; [host] is host memory.  The library never allocates or frees device
 *    memory, never synchronises: it only enqueues kernels on `stream`
 *    (the reference's Metal paths force eval() before encoding,
 *    overlap_add.cpp:45-48 — not reproduced).
 *  - all arrays are contiguous float32 (complex = interleaved re,im float32),
 *    row-major, shapes as in the reference's Python API.
 *  - return value: 0 = ok; AP_ERR_INVALID (<0) = invalid argument (the
 *    reference throws std::invalid_argument -> Python ValueError, e.g.
 *    overlap_add.cpp:205-222); AP_ERR_UNSUPPORTED; AP_ERR_HIP = a HIP runtime
 *    error at launch.  ap_last_error() returns a thread-local message.
 *  - re-entrant, no hidden state: windows, filterbanks, twiddles and FIR taps
 *    are inputs (built on the host in float64 by the ap_*_host builders below,
 *    cached per device by the caller).
 */
#ifndef AUDIOPRIMS_H
#define AUDIOPRIMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AP_OK 0
#define AP_ERR_INVALID (-1)
#define AP_ERR_UNSUPPORTED (-2)
#define AP_ERR_HIP (-3)

/* pad modes — stft.py:441-468, pad_signal.metal:10-121 */
#define AP_PAD_CONSTANT 0
#define AP_PAD_EDGE 1
#define AP_PAD_REFLECT 2

/* window kinds — windows.cpp:179-228 */
#define AP_WIN_HANN 0
#define AP_WIN_HAMMING 1
#define AP_WIN_BLACKMAN 2
#define AP_WIN_BARTLETT 3
#define AP_WIN_RECTANGULAR 4

int ap_version(void);
const char *ap_last_error(void);

/* ---------------------------------------------------------------------- *
 * Host-side builders (float64 internally).  Replace the CPU-stream parts of
 * the reference extension.
 * ---------------------------------------------------------------------- */

/* generate_window(window_type, length, periodic) — bindings.cpp:105-130,
 * windows.cpp:179-228: float64 build, cast to float32, average with own
 * reverse (windows.cpp:73-78); periodic = (length+1)-point window minus the
 * last point.  out_host: `length` floats. */
int ap_generate_window_host(int kind, int length, int periodic, float *out_host);

/* hz_to_mel / mel_to_hz(arr, htk) — bindings.cpp:133-181, mel_filterbank.cpp:70-106.
 * float64 in/out on the host. */
int ap_hz_to_mel_host(const double *hz, int64_t n, int htk, double *out);
int ap_mel_to_hz_host(const double *mel, int64_t n, int htk, double *out);

/* mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk, norm) — bindings.cpp:183-221,
 * mel_filterbank.cpp:108-239 (librosa's fdiff/ramps formulation, float64, one cast).
 * fmax < 0 means sr/2.  norm_slaney: 1 = "slaney", 0 = none.
 * out_host: n_mels*(n_fft/2+1) floats. */
int ap_mel_filterbank_host(int sr, int n_fft, int n_mels, double fmin, double fmax,
                           int htk, int norm_slaney, float *out_host);

/* get_dct_matrix(n_out, n_in, norm) — bindings.cpp:313-335, dct.cpp:24-101:
 * DCT-II basis evaluated in FLOAT32 like the reference's native path (dct.cpp:59).
 * out_host: n_out*n_in floats. */
int ap_dct_matrix_host(int n_out, int n_in, int ortho, float *out_host);

/* Twiddle table for an n_fft-point real transform: out_host[2j] = cos(2*pi*j/n),
 * out_host[2j+1] = sin(2*pi*j/n), j in [0,n), float64-evaluated, exact at the
 * quadrant points.  (Stands in for whatever mx.fft precomputes internally.) */
int ap_twiddle_table_host(int n_fft, float *out_host);

/* FIR of scipy.signal.resample_poly for the reduced ratio up/down (the reference's
 * resample_poly is exactly that SciPy call, resample.py:279-281):
 *   firwin(2*10*max(up,down)+1, 1/max(up,down), window=('kaiser', 5.0)) cast to float32,
 *   times up (float32), with n_pre_pad = down - half_len % down zeros prepended.
 * out_host must hold ap_resample_poly_ntaps(up, down) floats; n_pre_remove_out gets
 * (half_len + n_pre_pad) / down, the number of leading outputs SciPy discards. */
int ap_resample_poly_ntaps(int up, int down);
int ap_resample_poly_taps_host(int up, int down, float *out_host, int *n_pre_remove_out);

/* 1 if the LDS FFT engine can transform n_fft-point real frames, else 0. */
int ap_fft_supported(int n_fft);

/* ---------------------------------------------------------------------- *
 * Device primitives — 1:1 with the reference extension's hot-path entries.
 * ---------------------------------------------------------------------- */

/* pad_signal(signal (B,L), pad_length, mode) -> (B, L+2*pad) — bindings.cpp:76-102,
 * pad_signal.cpp:133-162.  reflect requires pad <= L-1 (pad_signal.cpp:102-105). */
int ap_pad_f32(const float *x /*dev*/, int64_t B, int64_t L, int64_t pad, int mode,
               float *out /*dev (B, L+2*pad)*/, void *stream);

/* frame_signal(signal (B,L), frame_length, hop) -> (B,T,frame_length),
 * T = 1 + (L-frame_length)/hop — bindings.cpp:48-74, frame_signal.cpp:119-155. */
int ap_frame_f32(const float *x /*dev*/, int64_t B, int64_t L, int frame_length, int hop,
                 float *out /*dev (B,T,frame_length)*/, void *stream);

/* overlap_add(frames (B,T,N), window (N,), hop, output_length) -> (B,out_len)
 * — bindings.cpp:15-46, overlap_add.cpp:195-233, overlap_add.metal:16-55:
 *   out[b,i] = sum_f w[p-fH]*frames[b,f,p-fH] / max(sum_f w[p-fH]^2, 1e-8),  p = i + out_offset
 * `out_offset` (>=0) lets istft fold its centre trim (stft.py:315-329) into the
 * same pass; the reference entry is out_offset = 0. */
int ap_overlap_add_f32(const float *frames /*dev*/, const float *window /*dev*/, int64_t B,
                       int64_t T, int n_fft, int hop, int64_t out_offset, int64_t out_len,
                       float *out /*dev (B,out_len)*/, void *stream);

/* ---------------------------------------------------------------------- *
 * Fused transforms — replace `_stft_core` (pad -> frame -> *window -> mx.fft.rfft,
 * stft.py:109-133) plus the transpose (stft.py:216), and mel.py:310-350.
 * ---------------------------------------------------------------------- */

/* stft: y (B,L) -> out (B, F=n_fft/2+1, T) complex64 (interleaved).
 *   window : n_fft floats, already centre-padded (stft.py:88-106)
 *   tw     : table from ap_twiddle_table_host(n_fft) copied to the device
 *   center : pad n_fft/2 both sides with `pad_mode` without materialising it
 *   T must equal 1 + (L + 2*pad - n_fft)/hop (_frame_impl.py:61). */
int ap_stft_f32(const float *y /*dev*/, int64_t B, int64_t L, int n_fft, int hop,
                const float *window /*dev*/, const float *tw /*dev*/, int center, int pad_mode,
                int64_t T, float *out /*dev (B,F,T,2)*/, void *stream);

/* The same with the rows of `out` `row_stride` complex values apart (row_stride >= T; the (B, F, T)
 * result is then a strided view of a (B, F, row_stride) buffer and the columns T..row_stride-1 are
 * never written).  With row_stride a multiple of 16 and a 128-byte aligned `out` every group of 16
 * frames leaves as whole 128-byte lines - the workspace layout of the Griffin-Lim loop
 * (griffinlim.py:129-180), where the spectra never leave the library.  No reference counterpart:
 * mx.fft.rfft + transpose (stft.py:130,216) always yields the dense layout, which is row_stride = T. */
int ap_stft_rows_f32(const float *y /*dev*/, int64_t B, int64_t L, int n_fft, int hop,
                     const float *window /*dev*/, const float *tw /*dev*/, int center, int pad_mode,
                     int64_t T, int64_t row_stride, float *out /*dev (B,F,row_stride,2)*/, void *stream);

/* Mel contraction plan.  The filterbank is tiny and built on the host (mel.py:100-168),
 * so its sparsity is analysed on the host once and shipped to the device next to it.
 * ap_mel_plan_host fills a device-bound blob `plan` (int32 words, copy it to HBM) and a
 * 16-int host descriptor `desc` that the caller hands back to ap_melspec_f32:
 *   desc[0] flags   AP_PLAN_BANDED | AP_PLAN_PARTS      desc[1] n_mels   desc[2] n_bins
 *   desc[3] total words of the blob
 *   desc[4] off band_lo[M]   desc[5] off band_len[M]    (span holding all non-zeros of a filter)
 *   desc[6] off parts[n_parts][4] = (slot, first 4-bin group, n_groups<=4, first quad)
 *   desc[7] n_parts          desc[8] off quads[n_quads][4] float weights   desc[9] n_quads
 *   desc[10] off rowstart[M+1]: row m's partial sums live in slots [rowstart[m], rowstart[m+1])
 *   desc[11] off wave_parts[n_wave_parts][4]: the same parts dealt to wavefront lanes (entry
 *            64 p + l = lane l of pass p) as (slot A, group A, slot B, group B): the entry's
 *            weight rows 0-1 go with |X|^p groups A, A+1 and sum into slot A, rows 2-3 go with
 *            groups B, B+1 and sum into slot B - a second part of <= 2 groups - or, slot B = -1,
 *            on top of slot A (one part of 3-4 groups), or into the dump slot n_parts (nothing
 *            there; an idle lane has both slots = n_parts).  Entries are ordered so that the
 *            16-byte LDS reads of one pass fall into different banks
 *   desc[13] off wave_quads[n_wave_quads][4]: lane-interleaved weights, row i of entry 64 p + l at
 *            256 p + 64 i + l (zero where a part ends)
 *   desc[12] n_wave_parts (multiple of 64)   desc[14] n_wave_quads = 4 n_wave_parts
 *   desc[15] largest number of parts of one row
 * Parts split each filter's span into runs of <= 4 aligned 4-bin groups, sorted by
 * length, so the wave kernel can contract with 16-byte LDS reads.  Zeros outside a
 * span contribute exactly 0 to the reference's matmul (mel.py:344-350), so using the
 * plan does not change which products are summed. */
#define AP_PLAN_BANDED 1
#define AP_PLAN_PARTS 2
#define AP_PLAN_FORCE_GENERIC 256   /* caller-set in desc[0]: keep the generic LDS engine (tests) */
#define AP_PLAN_DESC_INTS 16
int64_t ap_mel_plan_words(const float *fb_host /*(M,F)*/, int n_mels, int n_bins);
int ap_mel_plan_host(const float *fb_host /*(M,F)*/, int n_mels, int n_bins,
                     int32_t *plan_host, int32_t *desc_host /*16 ints*/);

/* melspectrogram: y (B,L) -> out (B, n_mels, T) = fb @ |stft(y)|^power, fused
 * (no (B,F,T) intermediate).  fb is the dense (n_mels, F) filterbank
 * (mel.py:100-168); plan (device) + desc (host) come from ap_mel_plan_host, both may
 * be NULL: dense contraction.  n_fft = 2048 with a PARTS plan runs the wave-per-frame
 * kernel (kernels_wave.h); everything else the generic LDS engine (kernels_generic.h). */
int ap_melspec_f32(const float *y /*dev*/, int64_t B, int64_t L, int n_fft, int hop,
                   const float *window /*dev*/, const float *tw /*dev*/, int center,
                   int pad_mode, int64_t T, const float *fb /*dev (M,F)*/,
                   const int32_t *plan /*dev*/, const int32_t *desc /*host, 16 ints*/,
                   int n_mels, float power, float *out /*dev (B,M,T)*/, void *stream);

/* The same, and also raises *max_key_dev (order-preserving key as in ap_reduce_max_f32, reset by
 * the call) to max(out): mfcc's top_db clip needs the global maximum of the mel power, and the
 * n_fft = 2048 kernel gets it for one atomic per wavefront instead of another pass over out. */
int ap_melspec_max_f32(const float *y /*dev*/, int64_t B, int64_t L, int n_fft, int hop,
                       const float *window /*dev*/, const float *tw /*dev*/, int center, int pad_mode,
                       int64_t T, const float *fb /*dev*/, const int32_t *plan /*dev or NULL*/,
                       const int32_t *desc /*host or NULL*/, int n_mels, float power,
                       float *out /*dev*/, uint32_t *max_key_dev /*dev or NULL*/, void *stream);
/* The same with the rows of `out` `row_stride` floats apart (row_stride >= T: the (B, M, T) result is a strided view of
 * a (B, M, row_stride) buffer, the columns T..row_stride-1 are never written).  With row_stride a multiple of 8 the
 * n_fft = 2048 run kernel's 8-frame output runs are whole aligned 32-byte sectors.  Served by that kernel only
 * (ap_melspec_rows_fused says whether it applies: n_fft = 2048, power 1 or 2, <= 128 filters, constant padding with
 * an even hop or center = 0); AP_ERR_UNSUPPORTED otherwise.  No reference counterpart: mx.matmul yields dense rows. */
int ap_melspec_rows_fused(int n_fft, int hop, int center, int pad_mode, int n_mels, float power,
                          const int32_t *plan /*host or NULL*/, const int32_t *desc /*host*/);
int ap_melspec_rows_f32(const float *y /*dev*/, int64_t B, int64_t L, int n_fft, int hop,
                        const float *window /*dev*/, const float *tw /*dev*/, int center, int pad_mode,
                        int64_t T, int64_t row_stride, const float *fb /*dev*/, const int32_t *plan /*dev*/,
                        const int32_t *desc /*host*/, int n_mels, float power, float *out /*dev (B,M,row_stride)*/,
                        uint32_t *max_key_dev /*dev or NULL*/, void *stream);

/* irfft of every frame: S (B,F,T) complex64 -> frames (B,T,n_fft) float32,
 * 1/n_fft scaled; imaginary parts of the DC and Nyquist bins are ignored —
 * mx.fft.irfft(·, n=n_fft) at stft.py:292-295 (incl. the transpose). */
int ap_irfft_frames_f32(const float *S /*dev (B,F,T,2)*/, int64_t B, int64_t T, int n_fft,
                        const float *tw /*dev*/, float *frames /*dev (B,T,n_fft)*/,
                        void *stream);

/* istft core: irfft + window + overlap-add + sum(w^2) normalise + trim.
 *   out[b,i] = OLA(position i + out_offset), i in [0,out_len)
 *   frames_ws : caller-provided workspace of ap_istft_workspace_floats(...) floats (B*T*n_fft
 *               in general; 0 — the pointer may then be NULL — when the fused n_fft = 2048
 *               kernel applies: the frames stay in LDS and never reach HBM).
 * Length logic (stft.py:300-338) stays in the caller. */
int64_t ap_istft_workspace_floats(int64_t B, int64_t T, int n_fft, int hop, int64_t out_offset);
int ap_istft_f32(const float *S /*dev (B,F,T,2)*/, int64_t B, int64_t T, int n_fft, int hop,
                 const float *window /*dev*/, const float *tw /*dev*/, float *frames_ws /*dev*/,
                 int64_t out_offset, int64_t out_len, float *out /*dev (B,out_len)*/,
                 void *stream);

/* The same from a spectrum whose rows are `row_stride` complex values apart (the layout ap_stft_rows_f32
 * writes; row_stride == T is the dense layout).  n_fft = 2048 with hop in {256, 512, 1024} and row_stride <= 490 000
 * (a clip is addressed as one buffer resource: 1025 row_stride 8 bytes < 0xF0000000), and the eight-frame sizes
 * (n_fft 512 / 400 / 256) only - the fused kernels, no workspace; AP_ERR_UNSUPPORTED otherwise - copy to a dense
 * array and call ap_istft_f32. */
int ap_istft_rows_f32(const float *S /*dev (B,F,row_stride,2)*/, int64_t B, int64_t T, int64_t row_stride,
                      int n_fft, int hop, const float *window /*dev*/, const float *tw /*dev*/,
                      int64_t out_offset, int64_t out_len, float *out /*dev (B,out_len)*/, void *stream);

/* resample_poly core: x (B,L) -> out (B, n_out), n_out = ceil(L*up/down),
 *   out[b,o] = sum_i taps[t - up*i] * x[b,i],  t = (o + n_pre_remove)*down,
 * float32 accumulation in increasing i like SciPy's upfirdn (padtype "constant").
 * up/down must already be gcd-reduced (resample.py:255-257). */
int ap_resample_poly_f32(const float *x /*dev*/, int64_t B, int64_t L, int up, int down,
                         const float *taps /*dev*/, int n_taps, int n_pre_remove,
                         int64_t n_out, float *out /*dev*/, void *stream);

/* scipy.signal.resample_poly's other padtypes (the reference forwards `padtype` to SciPy,
 * resample.py:279-281).  AP_EXT_* are scipy.signal.upfirdn's extension modes; ap_extend_f32 writes
 * (B, L + 2 n_ext): n_ext extension samples either side of each row (bit-identical to SciPy's
 * _extend_left / _extend_right).  ap_resample_poly_padded_f32 = extension into `ws`
 * (B * (L + 2 * ap_resample_poly_pad_samples(up, down, n_taps)) floats) + the polyphase filter; mode
 * AP_EXT_CONSTANT needs no workspace and is ap_resample_poly_f32.  'mean' / 'median' / 'minimum' /
 * 'maximum' are host-side: subtract the row statistic, filter with zeros outside, add it back. */
#define AP_EXT_CONSTANT 0
#define AP_EXT_WRAP 1
#define AP_EXT_EDGE 2
#define AP_EXT_SMOOTH 3
#define AP_EXT_SYMMETRIC 4
#define AP_EXT_REFLECT 5
#define AP_EXT_ANTISYMMETRIC 6
#define AP_EXT_ANTIREFLECT 7
#define AP_EXT_LINE 8
int ap_extend_f32(const float *x /*dev*/, int64_t B, int64_t L, int64_t n_ext, int mode,
                  float *out /*dev (B, L + 2 n_ext)*/, void *stream);
int64_t ap_resample_poly_pad_samples(int up, int down, int n_taps);
int ap_resample_poly_padded_f32(const float *x /*dev*/, int64_t B, int64_t L, int up, int down,
                                const float *taps /*dev*/, int n_taps, int n_pre_remove, int64_t n_out,
                                int mode, float *ws /*dev*/, float *out /*dev*/, void *stream);

/* linear-interpolation resampler (resample.py:142-212): positions j*(L-1)/(n_out-1)
 * evaluated in float64 like the reference's NumPy code. */
int ap_resample_linear_f32(const float *x /*dev*/, int64_t B, int64_t L, int64_t n_out,
                           double scale, float *out /*dev*/, void *stream);

/* FFT resampler = scipy.signal.resample(x, num) for real float32 rows (reference
 * resample.py:97,123): complex FFT of length Nx (four-step, both legs in LDS), SciPy's
 * Nyquist-aware spectrum truncation / zero-padding, inverse FFT of length num, scale num/Nx.
 *   ap_cfft_split_host(N, &N1, &N2): N = N1*N2 with both <= 4096, -1 if the length cannot be
 *     split (then the call below returns AP_ERR_UNSUPPORTED);
 *   tw_* : twiddle tables (ap_twiddle_table_host) of the four leg lengths on the device;
 *   ws   : workspace of 2 * B * max(Nx, num) complex64 (= 16 * B * max(Nx,num) bytes). */
int ap_cfft_split_host(int64_t N, int *N1, int *N2);
int ap_resample_fft_f32(const float *x /*dev (B,Nx)*/, int64_t B, int64_t Nx, int64_t num,
                        const float *tw_x1, const float *tw_x2, const float *tw_y1,
                        const float *tw_y2 /*dev*/, float *ws /*dev*/, float *out /*dev (B,num)*/,
                        void *stream);

/* The same for lengths with a prime factor > 4096: either transform may run as a chirp-z (Bluestein)
 * convolution of a supported length M >= 2 N - 1 (a power of two).  Mx / My = 0 keeps the direct
 * four-step transform for that side (tw_*: legs of N); Mx / My > 0: tw_* are the leg tables of M,
 * chirp_* (N complex64) = exp(-i pi n^2 / N), spec_* (M complex64) = FFT_M of conj(chirp) laid out
 * circularly (b[n] = b[M - n] = conj(chirp[n])), both built on the host in float64.
 *   ws : 2 * B * max(Nx, num, Mx, My) complex64. */
int ap_resample_fft_chirp_f32(const float *x /*dev (B,Nx)*/, int64_t B, int64_t Nx, int64_t num,
                              int64_t Mx, const float *tw_x1, const float *tw_x2,
                              const float *chirp_x, const float *spec_x,
                              int64_t My, const float *tw_y1, const float *tw_y2,
                              const float *chirp_y, const float *spec_y /*dev*/,
                              float *ws /*dev*/, float *out /*dev (B,num)*/, void *stream);

/* magnitude / phase / |S|^p of a complex64 array of n elements —
 * stft.py:347-379 (mx.abs, mx.arctan2). */
int ap_magnitude_f32(const float *S /*dev*/, int64_t n, float *out /*dev*/, void *stream);
int ap_phase_f32(const float *S /*dev*/, int64_t n, float *out /*dev*/, void *stream);
/* The same two from a spectrum whose rows are `row_stride` complex values apart (ap_stft_rows_f32) into a dense
 * (rows, T) float array: mode 0 = |S|, mode 1 = atan2(im, re). */
int ap_complex_unary_rows_f32(const float *S /*dev (rows,row_stride,2)*/, int64_t rows, int64_t T, int64_t row_stride,
                              int mode, float *out /*dev (rows,T)*/, void *stream);

/* ---------------------------------------------------------------------- *
 * Tails of the hot path: Griffin-Lim projection, dB conversion, DCT (mfcc).
 * ---------------------------------------------------------------------- */

/* Griffin-Lim (griffinlim.py:123-178), element-wise over (B*F, T):
 *   mode 0: rebuilt = S * exp(i*angles) (and tprev = rebuilt if tprev != NULL)
 *   mode 1: R' = S * exp(i*atan2(R.im, R.re)); rebuilt = R' + momentum*(R' - tprev);
 *           tprev = R' (when momentum > 0).  R has TR frames per row; frames >= TR
 *           count as zero (the reference crops / zero-pads R to T, :156-165). */
int ap_gl_project_f32(int mode, const float *S /*dev (BF,T)*/, const float *angles /*dev*/,
                      const float *R /*dev (BF,TR,2)*/, int64_t TR, int64_t BF, int64_t T,
                      float momentum, float *tprev /*dev (BF,T,2)*/, float *rebuilt /*dev*/,
                      void *stream);

/* out[i] = float32(low + (high-low) * next_double_i) of NumPy's PCG64 stream whose CURRENT state
 * is (state, inc) as reported by numpy.random.default_rng(seed).bit_generator.state, split in
 * 64-bit halves — i.e. exactly default_rng(seed).uniform(low, high, n).astype(float32)
 * (reference griffinlim.py:112-115), generated on the device. */
int ap_pcg64_uniform_f32(uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo,
                         double low, double high, int64_t n, float *out /*dev*/, void *stream);

/* The whole Griffin-Lim loop (griffinlim.py:123-191) enqueued by ONE call: rebuilt = S*exp(i*angles);
 * n_iter x { istft -> stft -> project+momentum }; final istft.  All buffers are the caller's:
 *   rebuilt, tprev : (B,F,T) complex64 scratch      R : (B,F,TR) complex64 scratch, TR = frames of stft(y)
 *   frames_ws : (B,T,n_fft) float32                 y : (B, y_len) float32 = the result
 * out_offset / y_len are istft's trim (see ap_istft_f32); TR must equal 1 + (y_len + 2*pad - n_fft)/hop. */
int ap_griffinlim_f32(const float *S /*dev (B,F,T)*/, const float *angles /*dev (B,F,T)*/, int64_t B,
                      int64_t T, int n_fft, int hop, const float *window /*dev*/,
                      const float *tw /*dev*/, int center, int pad_mode, int64_t out_offset,
                      int64_t y_len, int64_t TR, int n_iter, float momentum, float *rebuilt,
                      float *tprev, float *R, float *frames_ws, float *y /*dev*/, void *stream);

/* The same loop with the three complex workspaces held in rows `row_stride` complex values apart
 * ((B, F, row_stride) each; row_stride even, >= T, ideally a multiple of 16 with 128-byte aligned buffers so
 * that every 16-frame group is one whole line per row).  n_fft = 2048, hop in {256, 512, 1024}, TR == T (the
 * stft of the y_len-sample signal has as many frames as S) - AP_ERR_UNSUPPORTED otherwise: use
 * ap_griffinlim_f32.  S and angles are dense (B, F, T).  No frames workspace: the istft is the fused kernel. */
int ap_griffinlim_rows_f32(const float *S /*dev (B,F,T)*/, const float *angles /*dev (B,F,T)*/, int64_t B,
                           int64_t T, int64_t row_stride, int n_fft, int hop, const float *window /*dev*/,
                           const float *tw /*dev*/, int center, int pad_mode, int64_t out_offset,
                           int64_t y_len, int n_iter, float momentum, float *rebuilt, float *tprev, float *R,
                           float *y /*dev*/, void *stream);

/* out[0] = mean((a - b)^2) over n floats, deterministic (float64 partial sums per workgroup, added in a fixed
 * order): the reconstruction error griffinlim_iter returns (griffinlim.py:268-269).  ws: scratch of
 * ap_mse_workspace_doubles() float64 values. */
int64_t ap_mse_workspace_doubles(void);
int ap_mse_f32(const float *a /*dev*/, const float *b /*dev*/, int64_t n, double *ws /*dev*/, float *out /*dev*/,
               void *stream);

/* max over n floats into *key_dev (uint32 order-preserving key; caller provides the
 * 4-byte word, the call resets it first).  Used for ref=max and by ap_to_db_f32. */
int ap_reduce_max_f32(const float *x /*dev*/, int64_t n, uint32_t *key_dev, void *stream);

/* _to_db (convert.py:14-60): out = coef*log10(max(S,amin)/max(ref,amin)), then if
 * top_db >= 0: out = max(out, GLOBAL max(out) - top_db).  ref = *ref_key_dev (a key from
 * ap_reduce_max_f32) when ref_key_dev != NULL, else ref_value.  ws_dev: 4-byte scratch.
 * The conversion is monotone, so the clip floor is dB(max(S)) - top_db: one read-only max
 * reduction of S, then a single read+write pass. */
int ap_to_db_f32(const float *S /*dev*/, int64_t n, float coef, float amin, float ref_value,
                 const uint32_t *ref_key_dev, float top_db, float *out /*dev*/,
                 uint32_t *ws_dev, void *stream);

/* db_to_power (div=10) / db_to_amplitude (div=20): ref * 10^(x/div) — convert.py:100-198 */
int ap_from_db_f32(const float *x /*dev*/, int64_t n, float ref, float div, float *out /*dev*/,
                   void *stream);

/* dct(x, n, axis, norm) — bindings.cpp:337-368, dct.cpp:103-159, mfcc.py:69-140:
 *   x viewed as (outer, n_in, inner); out[o,k,i] = row_scale[k] * sum_m C[k,m] x[o,m,i];
 *   C is the (n_out, n_in) basis (ap_dct_matrix_host); row_scale may be NULL (lifter,
 *   mfcc.py:277-282). */
int ap_dct_f32(const float *x /*dev*/, const float *C /*dev (n_out,n_in)*/,
               const float *row_scale /*dev or NULL*/, int64_t outer, int n_in, int64_t inner,
               int n_out, float *out /*dev*/, void *stream);

/* mfcc tail (mfcc.py:253-287): power_to_db (+ top_db clip against the global maximum) fused into
 * the DCT's loads: out[o,k,i] = row_scale[k] * sum_m C[k,m] * dB(S[o,m,i]).  The dB array is
 * never materialised.  Arguments as in ap_to_db_f32 and ap_dct_f32; n_in * 64 (n_out <= 16) or
 * n_in * 128 bytes must fit 64 KiB of LDS (AP_ERR_UNSUPPORTED otherwise: use the two calls). */
int ap_db_dct_f32(const float *S /*dev*/, const float *C /*dev (n_out,n_in)*/,
                  const float *row_scale /*dev or NULL*/, int64_t outer, int n_in, int64_t inner,
                  int n_out, float coef, float amin, float ref_value, const uint32_t *ref_key_dev,
                  float top_db, uint32_t *ws_dev, int max_ready /* *ws_dev already holds max(S) */,
                  float *out /*dev*/, void *stream);

/* ---------------------------------------------------------------------- *
 * Callers that sit directly on the STFT / on the frames.
 * ---------------------------------------------------------------------- */

/* spectral_centroid / spectral_bandwidth / spectral_rolloff / spectral_flatness —
 * features.py:57-442, bindings.cpp:371-484 (spectral.cpp:8-257).  One pass over a spectrogram
 * S (B, F, T): real magnitudes (is_complex = 0) or the complex STFT (is_complex = 1: |X| is taken
 * on load); `power` raises it (features.py:52-54).  Outputs are (B, T) rows; NULL = not wanted.
 *   centroid  = sum(f S) / (sum S + 1e-10)
 *   bandwidth = (sum(S |f - c|^p) / (sum S + 1e-10))^(1/p)   (norm = 0: without the normaliser);
 *               c = centroid_in (B, T) when given, else the centroid above
 *   rolloff   = freq of the first bin whose running sum of S reaches roll_percent * sum S
 *   flatness  = exp(mean log max(S, amin)) / (mean max(S, amin) + 1e-10) */
int ap_spectral_stats_f32(const float *S /*dev*/, int is_complex, int64_t B, int64_t F, int64_t T,
                          const float *freq /*dev (F)*/, float power, const float *centroid_in /*dev or NULL*/,
                          float p, int norm, float roll_percent, float amin, float *centroid /*dev or NULL*/,
                          float *bandwidth, float *rolloff, float *flatness, void *stream);

/* spectral_contrast (reference features.py:445-595): bands (n_bands, 3) int32 on the device = first bin, one past
 * the last bin, number of extreme values k of every octave band (host-built from the bin frequencies as the
 * reference does); out (B, n_bands, T) = mean of the k largest minus mean of the k smallest magnitudes of the band,
 * as 10 log10 differences unless `linear`. */
int ap_spectral_contrast_f32(const float *S /*dev (B,F,T)*/, int64_t B, int64_t F, int64_t T,
                             const int32_t *bands /*dev*/, int n_bands, int linear, float *out /*dev*/, void *stream);

/* The same statistics straight from the audio for n_fft = 2048 (reference features.py:24-55: every
 * feature call runs its own STFT first): transform, |X|^power and the per-frame reductions in ONE kernel,
 * the complex spectrum never reaches HBM.  ap_spectral_audio_fused: 1 when the shape is served (n_fft
 * 2048, center=False or constant padding with an even hop); otherwise use ap_stft_f32 + ap_spectral_stats_f32.
 * Arguments as ap_stft_f32 + ap_spectral_stats_f32 (no centroid_in). */
int ap_spectral_audio_fused(int64_t L, int n_fft, int hop, int center, int pad_mode);
int ap_spectral_audio_f32(const float *y /*dev (B,L)*/, int64_t B, int64_t L, int n_fft, int hop,
                          const float *window, const float *tw, int center, int pad_mode, int64_t T,
                          const float *freq /*dev (n_fft/2+1)*/, float power, float p, int norm,
                          float roll_percent, float amin, float *centroid, float *bandwidth,
                          float *rolloff, float *flatness /*dev (B,T) or NULL*/, void *stream);

/* rms(y, frame_length, hop_length, center, pad_mode) and zero_crossing_rate(...) —
 * framing.py:81-150, features.py:598-722: per frame sqrt(mean x^2) and the fraction of samples
 * i >= 1 of the frame with (x[i] >= 0) != (x[i-1] >= 0).  pad = frame_length / 2 if center else 0,
 * pad_mode AP_PAD_CONSTANT or AP_PAD_EDGE; T = 1 + (L + 2 pad - frame_length) / hop.  Outputs (B, T). */
int ap_frame_stats_f32(const float *y /*dev (B,L)*/, int64_t B, int64_t L, int frame_length, int hop,
                       int center, int pad_mode, int64_t T, float *rms /*dev or NULL*/,
                       float *zcr /*dev or NULL*/, void *stream);

/* preemphasis(y, coef, zi) — framing.py:154-296: out[n] = y[n] - coef y[n-1], out[0] = y[0] + zi[b]
 * (zi == NULL: 2 y[0] - y[1]); zf[b] = y[L-1] (NULL = not wanted). */
int ap_preemphasis_f32(const float *y /*dev (B,L)*/, int64_t B, int64_t L, float coef,
                       const float *zi /*dev (B) or NULL*/, float *out /*dev*/, float *zf /*dev (B) or NULL*/,
                       void *stream);

/* deemphasis(y, coef, zi) — framing.py:298-392 (scipy.signal.lfilter([1], [1, -coef])):
 * out[n] = y[n] + coef out[n-1], out[0] = y[0] + zi[b].  zi == NULL: zero state and the reference's
 * correction ((2-coef) y[0] - y[1]) / (3-coef) * coef^n subtracted.  zf[b] = coef * out[L-1] of the
 * uncorrected recursion (lfilter's final state). */
int ap_deemphasis_f32(const float *y /*dev (B,L)*/, int64_t B, int64_t L, float coef,
                      const float *zi /*dev (B) or NULL*/, float *out /*dev*/, float *zf /*dev (B) or NULL*/,
                      void *stream);
/* The same with a workspace of ap_deemphasis_workspace_floats(B, L) floats: clips longer than 16 384
 * samples are filtered in chunks on workgroups of their own (chunk end states first, then every chunk
 * from the composed state it is entered with).  ws = NULL (or a short clip): one workgroup per clip. */
int64_t ap_deemphasis_workspace_floats(int64_t B, int64_t L);
int ap_deemphasis_ws_f32(const float *y /*dev*/, int64_t B, int64_t L, float coef, const float *zi /*dev (B) or NULL*/,
                         float *out /*dev*/, float *zf /*dev (B) or NULL*/, float *ws /*dev*/, void *stream);

/* delta(data, width, order, axis, mode) — mfcc.py:290-368 = scipy.signal.savgol_filter: FIR along the
 * middle axis of x viewed as (outer, n, inner); taps (width) in correlation order; mode AP_SG_*;
 * edge: (2 * (width/2), width) rows for mode AP_SG_INTERP (polynomial fit of the first / last
 * `width` samples), else NULL. */
#define AP_SG_INTERP 0
#define AP_SG_NEAREST 1
#define AP_SG_MIRROR 2
#define AP_SG_CONSTANT 3
#define AP_SG_WRAP 4
int ap_savgol_f32(const float *x /*dev*/, int64_t outer, int64_t n, int64_t inner, const float *taps /*dev*/,
                  int width, int mode, float cval, const float *edge /*dev or NULL*/, float *out /*dev*/,
                  void *stream);

/* 16-bit PCM ingest (the reference assumes float arrays already in memory,
 * README.md:34-38): out = x * scale (scale = 1/32768 for full-scale [-1, 1)). */
int ap_pcm16_to_f32(const int16_t *x /*dev*/, int64_t n, float scale, float *out /*dev*/, void *stream);

/* melspectrogram straight from 16-bit PCM: same arguments as ap_melspec_max_f32 with y (B, L) int16.
 * Where the n_fft = 2048 run kernel applies (ap_melspec_pcm16_fused() != 0: constant padding or
 * center = 0, power 2, n_mels <= 128, even L / hop) the conversion rides on the kernel's sample
 * loads (half the HBM read bytes, no float copy of the batch); otherwise the samples are first
 * converted into scratch_f32 (B*L floats, required then) and the float path runs. */
int ap_melspec_pcm16_fused(int64_t L, int n_fft, int hop, int center, int pad_mode, int n_mels, float power,
                           const int32_t *desc /*host*/);
int ap_melspec_pcm16_f32(const int16_t *y /*dev (B,L)*/, int64_t B, int64_t L, int n_fft, int hop,
                         const float *window /*dev*/, const float *tw /*dev*/, int center, int pad_mode,
                         int64_t T, const float *fb /*dev*/, const int32_t *plan /*dev*/,
                         const int32_t *desc /*host*/, int n_mels, float power, float *out /*dev*/,
                         uint32_t *max_key /*dev or NULL*/, float *scratch_f32 /*dev (B,L) or NULL*/, void *stream);

/* autocorrelation(y, max_lag, normalize, center) — pitch.py:16-115, bindings.cpp:224-252
 * (autocorrelation.cpp:10-84): r = irfft(|rfft(y - mean(y), n_fft)|^2)[:max_lag], divided by
 * max(r[0], 1e-10) when normalize; n_fft = ap_autocorrelation_nfft(n) = the next power of two
 * >= 2 n - 1 = N1 * N2 (ap_cfft_split_host).  tw1 / tw2: twiddle tables of the two legs
 * (ap_twiddle_table_host(N1), (N2)); ws: 4 B n_fft + B floats of scratch; out: (B, max_lag). */
int64_t ap_autocorrelation_nfft(int64_t n);
int ap_autocorrelation_f32(const float *y /*dev (B,n)*/, int64_t B, int64_t n, int64_t max_lag, int normalize,
                           int center, const float *tw1 /*dev*/, const float *tw2 /*dev*/, float *ws /*dev*/,
                           float *out /*dev*/, void *stream);

/* pitch_detect_acf / periodicity (reference pitch.py:118-369) from the raw autocorrelation r (rows, n_lag) of
 * every centred frame: on r / r[0] over the lags min_lag .. max_lag the first local maximum above `threshold`
 * (else the global maximum, if above it) gives f0 = sr / lag and voiced = 1; periodicity = the maximum.
 * Rows with r[0] <= 1e-10 stay 0.  Any of the three outputs may be NULL. */
int ap_acf_peaks_f32(const float *r /*dev*/, int64_t rows, int n_lag, int min_lag, int max_lag,
                     float threshold, float sr, float *f0 /*dev (rows)*/,
                     unsigned char *voiced /*dev (rows)*/, float *periodicity /*dev (rows)*/, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOPRIMS_H */
