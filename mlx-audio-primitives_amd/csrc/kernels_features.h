// Callers that sit directly on the STFT or on the frames: spectral
// centroid / bandwidth / rolloff / flatness over a (B, F, T) spectrogram, RMS and zero-crossing
// rate over the frames of a signal, pre- / de-emphasis and the Savitzky-Golay delta filter.
// All of them are HBM-streaming kernels: every input element is read from HBM once (second passes
// hit L2 / LDS), outputs are (B, 1, T) rows or arrays of the input's size.
//
// Reference: features.py:57-442,598-722; framing.py:81-392; mfcc.py:290-368;
// native mirrors csrc/primitives/spectral.cpp:8-257.
#pragma once
#include "kernels_generic.h"

// ---------------------------------------------------------------------------------------
// Spectral statistics of every frame of S (B, F, T) (T fastest, librosa layout), real magnitudes
// or complex STFT output (then |X| is taken on load; `power` != 1 raises it, features.py:52-54).
//
// A workgroup owns 32 consecutive frames of one clip: thread (tx = frame, ty = one of 8 contiguous
// bin stripes).  A wave reads two rows x 32 frames = 2 x 128 contiguous bytes per instruction.
//   pass A: per stripe  sum S, sum f S, sum log(max(S, amin)), sum max(S, amin)   -> LDS -> totals
//   pass B: sum S |f - centroid|^p per stripe; the stripe whose running sum crosses
//           roll_percent * total rescans its bins for the first one at or above it (rows come from L2)
struct ApSpectralParams {
    const float *S;            // (B, F, T) real, or (B, F, T, 2) complex
    const float *freq;         // (F)
    const float *centroid_in;  // optional (B, T): bandwidth around a given centroid
    float *centroid, *bandwidth, *rolloff, *flatness;      // (B, T) each, any may be NULL
    int64_t F, T, tiles_per_clip;
    int is_complex, norm;
    float power, p, roll_percent, amin;
};

#define APF_TX 32
#define APF_TY 8

AP_DEV float apf_load_mag(const ApSpectralParams &P, const float *Sb, int64_t k, int64_t t) {
    float v;
    if (P.is_complex) {
        const ap_float2 z = reinterpret_cast<const ap_float2 *>(Sb)[k * P.T + t];
        v = sqrtf(z.x * z.x + z.y * z.y);
    } else {
        v = Sb[k * P.T + t];
    }
    if (P.power != 1.0f) v = P.power == 2.0f ? v * v : powf(v, P.power);
    return v;
}

__global__ void __launch_bounds__(APF_TX * APF_TY) ap_spectral_stats_kernel(ApSpectralParams P) {
    __shared__ float red[5][APF_TY][APF_TX];
    const int tx = threadIdx.x & (APF_TX - 1), ty = threadIdx.x / APF_TX;
    const int64_t b = blockIdx.x / P.tiles_per_clip;
    const int64_t t = (blockIdx.x - b * P.tiles_per_clip) * APF_TX + tx;
    const bool live = t < P.T;
    const float *Sb = P.S + b * P.F * P.T * (P.is_complex ? 2 : 1);
    const int64_t per = (P.F + APF_TY - 1) / APF_TY;
    const int64_t k0 = ty * per, k1 = k0 + per < P.F ? k0 + per : P.F;

    float s0 = 0.0f, s1 = 0.0f, sl = 0.0f, sa = 0.0f;
    const bool want_flat = P.flatness != nullptr;            // 1025 logs per frame only when they are wanted
    if (live) {
        int64_t k = k0;
        for (; k + 4 <= k1; k += 4) {                        // four rows in flight per thread
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = apf_load_mag(P, Sb, k + u, t);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 += v[u];
                s1 = fmaf(P.freq[k + u], v[u], s1);
                if (want_flat) {
                    const float c = fmaxf(v[u], P.amin);
                    sl += logf(c);
                    sa += c;
                }
            }
        }
        for (; k < k1; ++k) {
            const float v = apf_load_mag(P, Sb, k, t);
            s0 += v;
            s1 = fmaf(P.freq[k], v, s1);
            if (want_flat) {
                const float c = fmaxf(v, P.amin);
                sl += logf(c);
                sa += c;
            }
        }
    }
    red[0][ty][tx] = s0; red[1][ty][tx] = s1; red[2][ty][tx] = sl; red[3][ty][tx] = sa;
    __syncthreads();
    float tot = 0.0f, tf = 0.0f, tl = 0.0f, ta = 0.0f, before = 0.0f;
#pragma unroll
    for (int j = 0; j < APF_TY; ++j) {
        if (j == ty) before = tot;             // running sum of S at the start of this stripe
        tot += red[0][j][tx]; tf += red[1][j][tx]; tl += red[2][j][tx]; ta += red[3][j][tx];
    }
    const float cen = P.centroid_in ? (live ? P.centroid_in[b * P.T + t] : 0.0f) : tf / (tot + 1e-10f);
    if (live && ty == 0) {
        if (P.centroid) P.centroid[b * P.T + t] = tf / (tot + 1e-10f);
        if (P.flatness) P.flatness[b * P.T + t] = expf(tl / (float)P.F) / (ta / (float)P.F + 1e-10f);
    }
    if (!P.bandwidth && !P.rolloff) return;
    // pass B
    const float thr = tot * P.roll_percent;
    const bool mine = live && P.rolloff && (before + s0 >= thr) && (ty == 0 || before < thr);   // first stripe that reaches thr
    float dev = 0.0f, run = before;
    int64_t found = -1;
    if (live && (P.bandwidth || mine)) {
        int64_t k = k0;
        for (; k + 4 <= k1; k += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = apf_load_mag(P, Sb, k + u, t);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (P.bandwidth) {
                    const float d = fabsf(P.freq[k + u] - cen);
                    dev = fmaf(v[u], P.p == 2.0f ? d * d : powf(d, P.p), dev);
                }
                run += v[u];
                if (mine && found < 0 && run >= thr) found = k + u;
            }
        }
        for (; k < k1; ++k) {
            const float v = apf_load_mag(P, Sb, k, t);
            if (P.bandwidth) {
                const float d = fabsf(P.freq[k] - cen);
                dev = fmaf(v, P.p == 2.0f ? d * d : powf(d, P.p), dev);
            }
            run += v;
            if (mine && found < 0 && run >= thr) found = k;
        }
    }
    if (mine) P.rolloff[b * P.T + t] = P.freq[found < 0 ? k1 - 1 : found];
    // a frame whose sums never reach thr (NaNs) keeps argmax's answer: bin 0
    if (live && P.rolloff && ty == 0 && !(tot >= thr)) P.rolloff[b * P.T + t] = P.freq[0];
    if (P.bandwidth) {
        red[4][ty][tx] = dev;
        __syncthreads();
        if (live && ty == 0) {
            float w = 0.0f;
#pragma unroll
            for (int j = 0; j < APF_TY; ++j) w += red[4][j][tx];
            if (P.norm) w = w / (tot + 1e-10f);
            P.bandwidth[b * P.T + t] = P.p == 2.0f ? sqrtf(w) : powf(w, 1.0f / P.p);
        }
    }
}

// ---------------------------------------------------------------------------------------
// RMS energy and zero-crossing rate of every frame (framing.py:81-150, features.py:598-722):
// frames of frame_length samples every hop, constant / edge padding by `pad` on both sides.
// A workgroup stages the contiguous span of G frames in LDS once (pad remap in the loader); wave w
// then reduces frames w, w + 4, ... with 64 lanes striding the frame.
//   zcr: a crossing at sample i >= 1 of the frame when (x[i] >= 0) != (x[i-1] >= 0); mean over
//        frame_length (the first sample never counts)
struct ApFrameStatsParams {
    const float *y;            // (B, L)
    float *rms, *zcr;          // (B, T), either may be NULL
    int64_t L, T, tiles_per_clip;
    int frame_length, hop, pad, pad_mode, G;
};

__global__ void __launch_bounds__(AP_BLOCK) ap_frame_stats_kernel(ApFrameStatsParams P) {
    float *span = reinterpret_cast<float *>(ap_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x / P.tiles_per_clip;
    const int64_t t0 = (blockIdx.x - b * P.tiles_per_clip) * P.G;
    const int Gt = (int)((P.T - t0) < P.G ? (P.T - t0) : P.G);
    const float *yb = P.y + b * P.L;
    const int64_t base = t0 * P.hop - P.pad;
    const int n_span = (Gt - 1) * P.hop + P.frame_length;
    for (int i = tid; i < n_span; i += AP_BLOCK) span[i] = ap_load_padded(yb, P.L, base + i, P.pad_mode);
    __syncthreads();
    for (int g = wave; g < Gt; g += AP_BLOCK / 64) {
        const float *fr = span + g * P.hop;
        float ss = 0.0f;
        int nc = 0;
        for (int i = lane; i < P.frame_length; i += 64) {
            const float x = fr[i];
            ss = fmaf(x, x, ss);
            if (i > 0) nc += ((x >= 0.0f) != (fr[i - 1] >= 0.0f)) ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ss += __shfl_xor(ss, off, 64);
            nc += __shfl_xor(nc, off, 64);
        }
        if (lane == 0) {
            if (P.rms) P.rms[b * P.T + t0 + g] = sqrtf(ss / (float)P.frame_length);
            if (P.zcr) P.zcr[b * P.T + t0 + g] = (float)nc / (float)P.frame_length;
        }
    }
}

// The same for frame_length = m hop (the usual 2048 / 512, 1024 / 256, 400 / 160 ... shapes): every sample
// is read ONCE.  A wave reduces one hop-sized block of the padded signal (16-byte loads, 4 samples per
// lane and load) to three numbers - sum of squares, sign changes inside the block, sign change across its
// left edge - and a frame is the sum of its m blocks (all inner edges, not its own left edge: the first
// sample of a frame never counts, features.py:700-716).  A workgroup covers G frames = G + m - 1 blocks.
struct ApFrameBlocksParams {
    const float *y;            // (B, L)
    float *rms, *zcr;          // (B, T), either may be NULL
    int64_t L, T, tiles_per_clip;
    int frame_length, hop, pad, pad_mode, m, G;
};
#define APF_MAX_BLOCKS 256

AP_DEV int apf_cross(float a, float b) { return ((a >= 0.0f) != (b >= 0.0f)) ? 1 : 0; }

__global__ void __launch_bounds__(AP_BLOCK) ap_frame_stats_blocks_kernel(ApFrameBlocksParams P) {
    __shared__ float bss[APF_MAX_BLOCKS];
    __shared__ int bzin[APF_MAX_BLOCKS], bzb[APF_MAX_BLOCKS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x / P.tiles_per_clip;
    const int64_t t0 = (blockIdx.x - b * P.tiles_per_clip) * P.G;
    const int Gt = (int)((P.T - t0) < P.G ? (P.T - t0) : P.G);
    const int nb = Gt + P.m - 1;
    const float *yb = P.y + b * P.L;
    const int n4 = P.hop >> 2;                                // groups of 4 samples per block
    for (int jj = wave; jj < nb; jj += AP_BLOCK / 64) {
        const int64_t start = (t0 + jj) * (int64_t)P.hop - P.pad;        // unpadded index of the block's first sample
        const bool interior = start >= 1 && start + P.hop <= P.L;
        float ss = 0.0f;
        int zin = 0, zb = 0;
        float carry = interior ? yb[start - 1] : ap_load_padded(yb, P.L, start - 1, P.pad_mode);   // sample left of the block
        for (int g0 = 0; g0 < n4; g0 += 64) {
            const int gi = g0 + lane;
            const bool act = gi < n4;
            float x0 = 0.0f, x1 = 0.0f, x2 = 0.0f, x3 = 0.0f;
            if (act) {
                const int64_t s = start + 4 * (int64_t)gi;
                if (interior) {
                    const ap_rsp_f4u q = *reinterpret_cast<const ap_rsp_f4u *>(yb + s);
                    x0 = q.x; x1 = q.y; x2 = q.z; x3 = q.w;
                } else {
                    x0 = ap_load_padded(yb, P.L, s, P.pad_mode);
                    x1 = ap_load_padded(yb, P.L, s + 1, P.pad_mode);
                    x2 = ap_load_padded(yb, P.L, s + 2, P.pad_mode);
                    x3 = ap_load_padded(yb, P.L, s + 3, P.pad_mode);
                }
            }
            // the sample before x0: the previous lane's x3, or what the previous trip (the block's left
            // neighbour) left in `carry`
            float left = __shfl_up(x3, 1, 64);
            if (lane == 0) left = carry;
            if (act) {
                ss = fmaf(x0, x0, fmaf(x1, x1, fmaf(x2, x2, fmaf(x3, x3, ss))));
                const int c0 = apf_cross(left, x0);
                if (gi == 0) zb = c0; else zin += c0;
                zin += apf_cross(x0, x1) + apf_cross(x1, x2) + apf_cross(x2, x3);
            }
            carry = __shfl(x3, 63, 64);                       // last sample of this trip (n4 > 64 only)
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ss += __shfl_xor(ss, off, 64);
            zin += __shfl_xor(zin, off, 64);
            zb += __shfl_xor(zb, off, 64);
        }
        if (lane == 0) { bss[jj] = ss; bzin[jj] = zin; bzb[jj] = zb; }
    }
    __syncthreads();
    for (int g = tid; g < Gt; g += AP_BLOCK) {
        float ss = 0.0f;
        int nc = 0;
        for (int q = 0; q < P.m; ++q) {
            ss += bss[g + q];
            nc += bzin[g + q] + (q > 0 ? bzb[g + q] : 0);
        }
        if (P.rms) P.rms[b * P.T + t0 + g] = sqrtf(ss / (float)P.frame_length);
        if (P.zcr) P.zcr[b * P.T + t0 + g] = (float)nc / (float)P.frame_length;
    }
}

// ---------------------------------------------------------------------------------------
// Pre-emphasis (framing.py:154-296): out[n] = y[n] - coef y[n-1]; out[0] = y[0] + zi, zi = the
// caller's initial state or 2 y[0] - y[1]; zf = y[L-1].
__global__ void __launch_bounds__(AP_BLOCK)
ap_preemphasis_kernel(const float *y, int64_t B, int64_t L, float coef, const float *zi, float *out, float *zf) {
    const int64_t n = B * L, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int64_t b = e / L, i = e - b * L;
        const float x = y[e];
        float o;
        if (i == 0) {
            const float z = zi ? zi[b] : (L > 1 ? 2.0f * x - y[e + 1] : x);
            o = x + z;
        } else {
            o = x - coef * y[e - 1];
        }
        out[e] = o;
        if (zf && i == L - 1) zf[b] = x;
    }
}

// the same, four samples per thread as one 16-byte load and store (L % 4 == 0 and both buffers 16-byte aligned:
// a quad never straddles two clips), one index division per quad
__global__ void __launch_bounds__(AP_BLOCK)
ap_preemphasis4_kernel(const float *y, int64_t B, int64_t L, float coef, const float *zi, float *out, float *zf) {
    const int64_t n4 = B * L / 4, stride = (int64_t)gridDim.x * blockDim.x;
    const ap_float4 *y4 = reinterpret_cast<const ap_float4 *>(y);
    ap_float4 *o4 = reinterpret_cast<ap_float4 *>(out);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const int64_t e = 4 * q, b = e / L, i = e - b * L;
        const ap_float4 x = y4[q];
        ap_float4 o;
        if (i == 0) o.x = x.x + (zi ? zi[b] : 2.0f * x.x - x.y);
        else o.x = x.x - coef * y[e - 1];
        o.y = x.y - coef * x.x;
        o.z = x.z - coef * x.y;
        o.w = x.w - coef * x.z;
        o4[q] = o;
        if (zf && i + 4 == L) zf[b] = x.w;
    }
}

// De-emphasis (framing.py:298-392): the recursion out[n] = y[n] + coef out[n-1] (scipy lfilter
// b = [1], a = [1, -coef]; out[0] = y[0] + zi), one workgroup per clip.  The clip is walked in tiles of
// 256 x 16 samples: a thread runs the recursion over its 16 consecutive samples from a zero state,
// the 256 end states are combined with the decay coef^16 per thread (Hillis-Steele scan of the affine
// maps s -> a s + b in LDS), then every thread adds carry * coef^(j+1) to its samples.
// librosa_zi != 0 (zi == NULL in the reference): zero initial state and the correction
// ((2-c) y0 - y1) / (3-c) * c^n subtracted from sample n.  zf = coef * (uncorrected) out[L-1].
//
// Long clips are cut into chunks of `chunk` samples (a multiple of the tile) that run on workgroups of their
// own: MODE 1 walks a chunk from a zero state and leaves only its end state Bv (the decay A = coef^len is
// the same for every full chunk); MODE 2 composes the maps of the chunks before its own
// (s -> A s + Bv, a handful of FMAs), then filters the chunk from that state.  12 bytes per sample instead
// of 8, but B x n_chunks workgroups instead of B.  MODE 0 = one workgroup per clip (short clips).
#define APD_PER 16
template <int MODE>
__global__ void __launch_bounds__(AP_BLOCK)
ap_deemphasis_kernel(const float *y, int64_t L, float coef, const float *zi, int librosa_zi, float *out, float *zf,
                     int64_t chunk, int n_chunks, float *carries) {
    __shared__ float tile[AP_BLOCK * (APD_PER + 1)];
    __shared__ float sa[2][AP_BLOCK], sb[2][AP_BLOCK];
    __shared__ float carry_s;
    const int tid = threadIdx.x;
    const int64_t b = MODE == 0 ? blockIdx.x : blockIdx.x / n_chunks;
    const int ck = MODE == 0 ? 0 : (int)(blockIdx.x - b * n_chunks);
    const int64_t lo = MODE == 0 ? 0 : ck * chunk;
    const int64_t hi = MODE == 0 ? L : (lo + chunk < L ? lo + chunk : L);
    const float *yb = y + b * L;
    float *ob = out + b * L;
    float cp[APD_PER + 1];                       // coef^j
    cp[0] = 1.0f;
#pragma unroll
    for (int j = 1; j <= APD_PER; ++j) cp[j] = cp[j - 1] * coef;
    const float corr = (MODE != 1 && librosa_zi && L > 1) ? ((2.0f - coef) * yb[0] - yb[1]) / (3.0f - coef) : 0.0f;
    if (tid == 0) {
        float s0 = librosa_zi ? 0.0f : (zi ? zi[b] : 0.0f);             // state entering sample 0: out[-1] * coef
        if (MODE == 1) s0 = 0.0f;
        if (MODE == 2) {
            const float A = powf(coef, (float)chunk);                    // every chunk before this one is full
            for (int c = 0; c < ck; ++c) s0 = fmaf(A, s0, carries[b * n_chunks + c]);
        }
        carry_s = s0;
    }
    __syncthreads();
    for (int64_t base = lo; base < hi; base += (int64_t)AP_BLOCK * APD_PER) {
        // coalesced load; thread tid owns samples [tid * 16, tid * 16 + 16) of the tile (rows padded by 1).
        // A whole tile at 16-byte-aligned addresses moves as float4 (the 4-byte form reaches 2.7 TB/s).
        const bool vec = base + (int64_t)AP_BLOCK * APD_PER <= hi &&
                         ((reinterpret_cast<uintptr_t>(yb + base) | reinterpret_cast<uintptr_t>(ob + base)) & 15) == 0;
        if (vec) {
            const ap_float4 *src = reinterpret_cast<const ap_float4 *>(yb + base);
#pragma unroll
            for (int q = 0; q < APD_PER / 4; ++q) {
                const int i4 = tid + q * AP_BLOCK;                  // float4 index in the tile: samples 4 i4 .. 4 i4 + 3
                const ap_float4 x4 = src[i4];
                float *row = tile + (i4 / (APD_PER / 4)) * (APD_PER + 1) + 4 * (i4 % (APD_PER / 4));
                row[0] = x4.x; row[1] = x4.y; row[2] = x4.z; row[3] = x4.w;
            }
        } else {
            for (int i = tid; i < AP_BLOCK * APD_PER; i += AP_BLOCK) {
                const int64_t n = base + i;
                tile[(i / APD_PER) * (APD_PER + 1) + (i % APD_PER)] = n < hi ? yb[n] : 0.0f;
            }
        }
        __syncthreads();
        float v[APD_PER];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < APD_PER; ++j) {
            s = fmaf(coef, s, tile[tid * (APD_PER + 1) + j]);       // local recursion from a zero state
            v[j] = s;
        }
        // affine map of this thread's segment: state_out = A state_in + Bv, A = coef^16, Bv = coef * s
        int cur = 0;
        sa[0][tid] = cp[APD_PER];
        sb[0][tid] = coef * s;
        __syncthreads();
        for (int off = 1; off < AP_BLOCK; off <<= 1) {              // inclusive scan of map composition
            float a = sa[cur][tid], bb = sb[cur][tid];
            if (tid >= off) {
                const float a0 = sa[cur][tid - off], b0 = sb[cur][tid - off];
                bb = fmaf(a, b0, bb);
                a = a * a0;
            }
            sa[cur ^ 1][tid] = a;
            sb[cur ^ 1][tid] = bb;
            cur ^= 1;
            __syncthreads();
        }
        const float cin = carry_s;                                   // coef * out[base - 1]
        // state entering this thread's segment = (exclusive prefix map)(cin)
        const float st = tid == 0 ? cin : fmaf(sa[cur][tid - 1], cin, sb[cur][tid - 1]);
        const float tile_out = fmaf(sa[cur][AP_BLOCK - 1], cin, sb[cur][AP_BLOCK - 1]);
        __syncthreads();
        if (tid == 0) carry_s = tile_out;
        if (MODE == 1) { __syncthreads(); continue; }                // only the chunk's end state is wanted
        // the correction's coef^n: one power per thread and tile, coef^j from the table (and nothing at all once
        // it has underflowed: coef^n is 0 in float32 a few thousand samples into the clip)
        const float pw = librosa_zi ? powf(coef, (float)(base + (int64_t)tid * APD_PER)) : 0.0f;
#pragma unroll
        for (int j = 0; j < APD_PER; ++j) {
            const int64_t n = base + (int64_t)tid * APD_PER + j;
            float o = fmaf(st, cp[j], v[j]);                         // + state * coef^j (state already holds one coef)
            if (zf && n == L - 1) zf[b] = coef * o;                  // lfilter's final state, before the correction
            if (librosa_zi) o -= corr * (pw * cp[j]);
            tile[tid * (APD_PER + 1) + j] = o;
        }
        __syncthreads();
        if (vec) {
            ap_float4 *dst = reinterpret_cast<ap_float4 *>(ob + base);
#pragma unroll
            for (int q = 0; q < APD_PER / 4; ++q) {
                const int i4 = tid + q * AP_BLOCK;
                const float *row = tile + (i4 / (APD_PER / 4)) * (APD_PER + 1) + 4 * (i4 % (APD_PER / 4));
                ap_float4 o4;
                o4.x = row[0]; o4.y = row[1]; o4.z = row[2]; o4.w = row[3];
                dst[i4] = o4;
            }
        } else {
            for (int i = tid; i < AP_BLOCK * APD_PER; i += AP_BLOCK) {
                const int64_t n = base + i;
                if (n < hi) ob[n] = tile[(i / APD_PER) * (APD_PER + 1) + (i % APD_PER)];
            }
        }
        __syncthreads();
    }
    if (MODE == 1 && tid == 0) carries[b * n_chunks + ck] = carry_s;
}

// ---------------------------------------------------------------------------------------
// Savitzky-Golay filter along the middle axis of x viewed as (outer, n, inner)
// (mfcc.py:290-368 -> scipy.signal.savgol_filter(deriv = order, delta = 1)):
//   out[o, i, r] = sum_j taps[j] x[o, remap(i + j - half), r]
// with scipy.ndimage's boundary modes for the remap (nearest / mirror / constant / wrap) and, for
// mode 'interp', the first / last `half` outputs taken from the polynomial fitted to the first / last
// `width` samples: edge[e][j] applied to those samples (host-built rows, e = 0..2 half - 1).
__global__ void __launch_bounds__(AP_BLOCK)
ap_savgol_kernel(const float *x, int64_t outer, int64_t n, int64_t inner, const float *taps, int width, int mode,
                 float cval, const float *edge, float *out) {
    const int half = width / 2;
    const int64_t total = outer * n * inner, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t o = e / (n * inner), rem = e - o * n * inner;
        const int64_t i = rem / inner, r = rem - i * inner;
        const float *xb = x + o * n * inner + r;
        float acc = 0.0f;
        if (mode == AP_SG_INTERP && (i < half || i >= n - half)) {
            const bool head = i < half;
            const float *row = edge + (head ? i : half + (i - (n - half))) * width;
            const int64_t first = head ? 0 : n - width;
            for (int j = 0; j < width; ++j) acc = fmaf(row[j], xb[(first + j) * inner], acc);
        } else {
            for (int j = 0; j < width; ++j) {
                int64_t q = i + j - half;
                float v;
                if (q >= 0 && q < n) {
                    v = xb[q * inner];
                } else if (mode == AP_SG_CONSTANT) {
                    v = cval;
                } else {
                    if (mode == AP_SG_NEAREST) q = q < 0 ? 0 : n - 1;
                    else if (mode == AP_SG_MIRROR) {                 // d c b | a b c d | c b a
                        if (n == 1) q = 0;
                        else {
                            const int64_t period = 2 * (n - 1);
                            q = ((q % period) + period) % period;
                            if (q >= n) q = period - q;
                        }
                    } else {                                         // wrap
                        q = ((q % n) + n) % n;
                    }
                    v = xb[q * inner];
                }
                acc = fmaf(taps[j], v, acc);
            }
        }
        out[e] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// Autocorrelation by the Wiener-Khinchin theorem (pitch.py:16-115, autocorrelation.cpp:10-84):
// r = irfft(|rfft(y - mean, n_fft)|^2)[:max_lag] / max(r[0], 1e-10), n_fft = next power of two
// >= 2 n - 1.  The two transforms are the four-step complex FFT of kernels_bigfft.h; these kernels
// are the glue around them.
__global__ void __launch_bounds__(AP_BLOCK) ap_row_mean_kernel(const float *y, int64_t n, float *mean) {
    __shared__ float red[AP_BLOCK];
    const float *yb = y + (int64_t)blockIdx.x * n;
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += AP_BLOCK) s += yb[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = AP_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[blockIdx.x] = red[0] / (float)n;
}

__global__ void __launch_bounds__(AP_BLOCK)
ap_autocorr_pad_kernel(const float *y, int64_t B, int64_t n, int64_t N, const float *mean, float *padded) {
    const int64_t total = B * N, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / N, i = e - b * N;
        padded[e] = i < n ? y[b * n + i] - (mean ? mean[b] : 0.0f) : 0.0f;
    }
}

__global__ void __launch_bounds__(AP_BLOCK) ap_power_spectrum_kernel(ap_float2 *X, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
        const ap_float2 z = X[e];
        X[e] = ap_mk(z.x * z.x + z.y * z.y, 0.0f);
    }
}

__global__ void __launch_bounds__(AP_BLOCK)
ap_autocorr_finish_kernel(const float *r, int64_t B, int64_t N, int64_t max_lag, int normalize, float *out) {
    const int64_t total = B * max_lag, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / max_lag, k = e - b * max_lag;
        float v = r[b * N + k];
        if (normalize) v = v / fmaxf(r[b * N], 1e-10f);
        out[e] = v;
    }
}

// ---------------------------------------------------------------------------------------
// 16-bit PCM -> float32 (x * scale, scale = 1 / 32768): the ingest step in front of the path
// (SURVEY.md §8f rank 3).  The n_fft = 2048 mel run kernel converts inside its sample loads instead;
// this pass serves every other shape.  8 samples (one 16-byte load, two 16-byte stores) per thread.
__global__ void __launch_bounds__(AP_BLOCK) ap_pcm16_to_f32_kernel(const int16_t *x, int64_t n, float scale, float *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n8 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? n / 8 : 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += stride) {
        const ap_int4 d = reinterpret_cast<const ap_int4 *>(x)[e];
        ap_float4 a, b;
        a.x = (float)(short)(d.x & 0xFFFF) * scale; a.y = (float)(d.x >> 16) * scale;
        a.z = (float)(short)(d.y & 0xFFFF) * scale; a.w = (float)(d.y >> 16) * scale;
        b.x = (float)(short)(d.z & 0xFFFF) * scale; b.y = (float)(d.z >> 16) * scale;
        b.z = (float)(short)(d.w & 0xFFFF) * scale; b.w = (float)(d.w >> 16) * scale;
        reinterpret_cast<ap_float4 *>(out)[2 * e] = a;
        reinterpret_cast<ap_float4 *>(out)[2 * e + 1] = b;
    }
    for (int64_t e = 8 * n8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride)
        out[e] = (float)x[e] * scale;
}

// The same along a contiguous axis (inner == 1: delta over time, the default axis = -1): a workgroup owns
// 1024 consecutive positions of one row (4 per thread), the taps sit in LDS, positions whose window lies
// inside the row take the plain FIR; the few at the row's ends go through the boundary rules above.  One
// 32-bit division per workgroup instead of two 64-bit ones per element.
#define APSG_PER 4
__global__ void __launch_bounds__(AP_BLOCK)
ap_savgol_rows_kernel(const float *x, int64_t rows, int n, int chunks, const float *taps, int width, int mode, float cval,
                      const float *edge, float *out) {
    __shared__ float tp[64];
    const int half = width / 2;
    if (threadIdx.x < width) tp[threadIdx.x] = taps[threadIdx.x];
    __syncthreads();
    const int64_t row = blockIdx.x / chunks;
    const int i0 = (int)(blockIdx.x - row * chunks) * (AP_BLOCK * APSG_PER);
    const float *xb = x + row * n;
    float *ob = out + row * n;
#pragma unroll
    for (int u = 0; u < APSG_PER; ++u) {
        const int i = i0 + u * AP_BLOCK + threadIdx.x;
        if (i >= n) break;
        float acc = 0.0f;
        if (i >= half && i < n - half) {
            const float *w = xb + i - half;
            for (int j = 0; j < width; ++j) acc = fmaf(tp[j], w[j], acc);
        } else if (mode == AP_SG_INTERP) {
            const bool head = i < half;
            const float *rw = edge + (head ? i : half + (i - (n - half))) * width;
            const int first = head ? 0 : n - width;
            for (int j = 0; j < width; ++j) acc = fmaf(rw[j], xb[first + j], acc);
        } else {
            for (int j = 0; j < width; ++j) {
                int q = i + j - half;
                float v;
                if (q >= 0 && q < n) v = xb[q];
                else if (mode == AP_SG_CONSTANT) v = cval;
                else {
                    if (mode == AP_SG_NEAREST) q = q < 0 ? 0 : n - 1;
                    else if (mode == AP_SG_MIRROR) {
                        if (n == 1) q = 0;
                        else {
                            const int period = 2 * (n - 1);
                            q = ((q % period) + period) % period;
                            if (q >= n) q = period - q;
                        }
                    } else q = ((q % n) + n) % n;
                    v = xb[q];
                }
                acc = fmaf(tp[j], v, acc);
            }
        }
        ob[i] = acc;
    }
}

// Signal extension of scipy.signal.upfirdn (the `mode` of scipy.signal.resample_poly's padtype, which the
// reference passes through at resample.py:279-281): out (B, L + 2 P) = P extension samples, the signal, P
// extension samples.  Every branch restates SciPy's _extend_left / _extend_right (third-party:
// scipy/signal/_upfirdn_apply.pyx, pinned by tests against scipy.signal._upfirdn_apply._pad_test) with the
// same float32 operation order and no fused multiply-add, so the extension is bit-identical.
#define AP_EXT_CONSTANT 0
#define AP_EXT_WRAP 1
#define AP_EXT_EDGE 2
#define AP_EXT_SMOOTH 3
#define AP_EXT_SYMMETRIC 4
#define AP_EXT_REFLECT 5
#define AP_EXT_ANTISYMMETRIC 6
#define AP_EXT_ANTIREFLECT 7
#define AP_EXT_LINE 8

// One rounding per operation: hipcc's default -ffp-contract=fast would fuse a product into the sum that
// follows it (and __fmul_rn / __fadd_rn are plain operators under it), so contraction is switched off here.
#ifdef AP_HOST_EMU
AP_DEV float ap_mul_rn(float a, float b) { volatile float r = a * b; return r; }
AP_DEV float ap_add_rn(float a, float b) { volatile float r = a + b; return r; }
#else
AP_DEV float ap_mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
AP_DEV float ap_add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
#endif
// float64 quotient rounded once more: exact for float32 operands (53 >= 2 * 24 + 2 bits), whatever the
// float32 divide expands to
AP_DEV float ap_div_rn(float a, float b) { return (float)((double)a / (double)b); }
AP_DEV float ap_sub_rn(float a, float b) { return ap_add_rn(a, -b); }

AP_DEV float ap_extend_sample(const float *x, int64_t L, int64_t i, int mode) {
    if (i >= 0 && i < L) return x[i];
    const bool left = i < 0;
    switch (mode) {
    case AP_EXT_WRAP: {
        int64_t j = i % L;
        if (j < 0) j += L;
        return x[j];
    }
    case AP_EXT_EDGE: return left ? x[0] : x[L - 1];
    case AP_EXT_SMOOTH:
        return left ? ap_add_rn(x[0], ap_mul_rn((float)i, ap_sub_rn(x[1], x[0])))
                    : ap_add_rn(x[L - 1], ap_mul_rn((float)(i - L + 1), ap_sub_rn(x[L - 1], x[L - 2])));
    case AP_EXT_LINE: {
        const float slope = ap_div_rn(ap_sub_rn(x[L - 1], x[0]), (float)(L - 1));
        return left ? ap_add_rn(x[0], ap_mul_rn((float)i, slope))
                    : ap_add_rn(x[L - 1], ap_mul_rn((float)(i - L + 1), slope));
    }
    case AP_EXT_SYMMETRIC:
    case AP_EXT_ANTISYMMETRIC: {
        int64_t j = i % (2 * L);
        if (j < 0) j += 2 * L;
        if (j < L) return x[j];
        const float v = x[2 * L - 1 - j];
        return mode == AP_EXT_ANTISYMMETRIC ? -v : v;
    }
    case AP_EXT_REFLECT: {
        const int64_t per = 2 * (L - 1);
        int64_t j = i % per;
        if (j < 0) j += per;
        return j < L ? x[j] : x[per - j];
    }
    case AP_EXT_ANTIREFLECT: {
        if (left) {
            if (-i < L) return ap_sub_rn(x[0], ap_sub_rn(x[-i], x[0]));
            const float le = ap_add_rn(x[0], ap_mul_rn(ap_sub_rn(x[0], x[L - 1]), (float)((-i - 1) / (L - 1))));
            const int64_t j = (-i - 1) % (2 * (L - 1));
            if (j < L - 1) return ap_sub_rn(le, ap_sub_rn(x[j + 1], x[0]));
            return ap_sub_rn(le, ap_sub_rn(x[L - 1], x[L - 2 - (j - (L - 1))]));
        }
        if (i < 2 * L - 1) return ap_sub_rn(x[L - 1], ap_sub_rn(x[2 * L - i - 2], x[L - 1]));
        const float re = ap_add_rn(x[L - 1], ap_mul_rn(ap_sub_rn(x[L - 1], x[0]), (float)(i / (L - 1) - 1)));
        const int64_t j = i % (2 * (L - 1));
        if (j < L - 1) return ap_add_rn(re, ap_sub_rn(x[j], x[0]));
        return ap_add_rn(re, ap_sub_rn(x[L - 1], x[2 * (L - 1) - j]));
    }
    default: return 0.0f;
    }
}

__global__ void __launch_bounds__(AP_BLOCK) ap_extend_kernel(const float *x, int64_t B, int64_t L, int64_t P, int mode,
                                                            float *out) {
    const int64_t Lo = L + 2 * P;
    const int64_t n = B * Lo;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int64_t b = e / Lo;
        const int64_t i = e - b * Lo - P;
        out[e] = ap_extend_sample(x + b * L, L, i, mode);
    }
}

// Pitch from the autocorrelation of every frame (reference pitch.py:118-369: pitch_detect_acf, periodicity):
// r (rows, n_lag) is the RAW autocorrelation of the centred frames (lags 0 .. n_lag - 1; lags past the frame
// are 0).  Per row, on r / r[0] over the lags min_lag .. max_lag: the first local maximum above `threshold`
// (else the global maximum if it is above it) gives f0 = sr / lag; the maximum itself is the periodicity.
// Rows with r[0] <= 1e-10 (silence) stay unvoiced / 0.  One thread per row.
__global__ void __launch_bounds__(AP_BLOCK)
ap_acf_peak_kernel(const float *r, int64_t rows, int n_lag, int min_lag, int max_lag, float threshold, float sr,
                   float *f0, unsigned char *voiced, float *periodicity) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const float *rr = r + row * n_lag;
    const float r0 = rr[0];
    float f = 0.0f, per = 0.0f;
    unsigned char v = 0;
    if (r0 > 1e-10f && max_lag >= min_lag) {
        auto at = [&](int lag) { return lag < n_lag ? rr[lag] / r0 : 0.0f; };
        int first = -1, arg = min_lag;
        float best = at(min_lag);
        float prev = best, cur = max_lag > min_lag ? at(min_lag + 1) : best;
        if (cur > best) { best = cur; arg = min_lag + 1; }
        for (int lag = min_lag + 1; lag < max_lag; ++lag) {            // interior points of the search range
            const float next = at(lag + 1);
            if (first < 0 && cur > prev && cur > next && cur > threshold) first = lag;
            if (next > best) { best = next; arg = lag + 1; }
            prev = cur;
            cur = next;
        }
        per = best;
        const int lag = first >= 0 ? first : (best > threshold ? arg : -1);
        if (lag > 0) { f = sr / (float)lag; v = 1; }
    }
    if (f0) f0[row] = f;
    if (voiced) voiced[row] = v;
    if (periodicity) periodicity[row] = per;
}

// Spectral contrast (reference features.py:445-595, librosa's rule): for every frame and octave band the mean of
// the k smallest and the mean of the k largest magnitudes of the band's bins [lo, hi); contrast = their
// difference, in dB unless `linear`.  The reference sorts every band on the host; k is a small fraction of the
// band (quantile 0.02: 1-9 values), so a thread extracts the k extremes by repeated selection instead - each
// pass finds the next element in (value, bin) order, so ties are taken once each - from rows that are coalesced
// across the frames of a workgroup and stay in L1 / L2 between passes.  One thread per (clip, frame), one
// grid row per band.
__global__ void __launch_bounds__(AP_BLOCK)
ap_spectral_contrast_kernel(const float *S, int64_t B, int64_t F, int64_t T, const int32_t *bands /* [n][3]: lo, hi, k */,
                            int linear, float *out /* (B, n, T) */) {
    const int band = blockIdx.y, n_bands = gridDim.y;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * T) return;
    const int64_t b = e / T, t = e - b * T;
    const int lo = bands[3 * band], hi = bands[3 * band + 1];
    int k = bands[3 * band + 2];
    float valley = 0.0f, peak = 0.0f;
    if (hi > lo) {
        if (k > hi - lo) k = hi - lo;
        const float *col = S + b * F * T + t;
        // k smallest, ascending: next element after (pv, pi) in (value, bin) order
        float pv = -INFINITY, sum = 0.0f;
        int pi = -1;
        for (int j = 0; j < k; ++j) {
            float bv = INFINITY;
            int bi = -1;
            for (int i = lo; i < hi; ++i) {
                const float v = col[(int64_t)i * T];
                const bool after = v > pv || (v == pv && i > pi);
                if (after && (v < bv || bi < 0)) { bv = v; bi = i; }
            }
            if (bi < 0) break;                     // NaNs: nothing comparable is left
            sum += bv;
            pv = bv;
            pi = bi;
        }
        valley = sum / (float)k;
        // k largest, descending
        pv = INFINITY;
        pi = hi;
        sum = 0.0f;
        for (int j = 0; j < k; ++j) {
            float bv = -INFINITY;
            int bi = -1;
            for (int i = hi - 1; i >= lo; --i) {
                const float v = col[(int64_t)i * T];
                const bool before = v < pv || (v == pv && i < pi);
                if (before && (v > bv || bi < 0)) { bv = v; bi = i; }
            }
            if (bi < 0) break;
            sum += bv;
            pv = bv;
            pi = bi;
        }
        peak = sum / (float)k;
    }
    float c;
    if (linear) c = peak - valley;
    else c = 10.0f * log10f(fmaxf(peak, 1e-10f)) - 10.0f * log10f(fmaxf(valley, 1e-10f));
    out[(b * n_bands + band) * T + t] = c;
}
