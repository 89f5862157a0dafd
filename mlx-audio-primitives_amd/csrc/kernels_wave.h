// Fused mel-spectrogram kernel for n_fft = 2048 on gfx950: one wavefront per frame.
//
// A 256-thread workgroup (4 wave64) owns a tile of 16 consecutive frames of one clip
// and walks it in 4 rounds; in a round every wave transforms ONE frame entirely in its
// own registers + a wave-private LDS exchange buffer (no s_barrier inside the FFT):
//
//   global samples (16 x 8 B per lane, prefetched one round ahead) * window (registers)
//   radix-16 butterflies in registers                              [mx.fft.rfft, stft.py:130]
//   * W_1024^(lane*k1) (registers) -> LDS transpose #1 (padded rows, ds_read_b128)
//   radix-16 butterflies, * W_64^(a*c) (small LDS table)
//   radix-4 across the 4 lanes of a quad with DPP quad_perm (no LDS round trip)
//   LDS transpose #2 to natural order -> paired real-input split (bins k and 1024-k
//   share their sums) -> |X|^p -> one float plane per wave in LDS
//
// then, once per round, the workgroup contracts the 4 planes with the mel filterbank
// (mx.matmul(mel_basis, S), mel.py:344-350) from a host-built plan: every filter's span
// is cut into parts of <= 4 aligned 4-bin groups; a thread owns (part, frame), does one
// ds_read_b128 of weights + one of |X|^p per group, and the <= 16 partial sums of a row are
// added by the thread that owns (row, frame).  No atomics (LDS float atomics cost ~3
// cycles per lane on gfx950 and made the first version of this kernel LDS-bound).
// After 4 rounds the (n_mels x 16) tile goes to HBM as 64-byte row segments of (B,M,T).
//
// The complex 1024-point transform is 16 x 16 x 4.  Workgroups are persistent over tiles.
#pragma once
#include "ap_wave_params.h"
#include "fft_lds.h"
#include "kernels_generic.h"

#ifdef AP_HOST_EMU
#define AP_WAVE_SYNC() emu_wave_sync()
#define AP_SCHED_FENCE() do {} while (0)
#else
#define AP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// wave-private LDS hand-off: the DS unit executes one wave's instructions in order, so
// only the compiler has to be kept from moving LDS accesses across this point.
#define AP_WAVE_SYNC()                                          \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

AP_DEV float ap_quad_xor1(float x) {   // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
AP_DEV float ap_quad_xor2(float x) {   // value of lane ^ 2 (quad_perm [2,3,0,1])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
}
#endif

struct __attribute__((aligned(16))) ap_float4 { float x, y, z, w; };
struct __attribute__((aligned(16))) ap_int4 { int x, y, z, w; };

// natural-order index k -> padded LDS slot: conflict-free ds_write_b64 of the quad
// outputs and (nearly) conflict-free ds_read_b64 of Z[lane+64r] / Z[1024-lane-64r]
// (tools/lds_banks.py)
AP_DEV int apw_zidx(int k) { return k + 4 * (k >> 8); }

// |X'|^p of X' = 2X with the exponent class fixed at compile time (2: power 2, 1: power 1,
// 0: anything else) so no powf code sits in the power-2 instruction stream.
template <int PMODE>
AP_DEV float apw_pow2x(float re, float im, float power) {
    const float p2 = re * re + im * im;
    if (PMODE == 2) return 0.25f * p2;
    if (PMODE == 1) return 0.5f * sqrtf(p2);
    return powf(0.5f * sqrtf(p2), power);
}

template <int PMODE>
__global__ void __launch_bounds__(256, 2) ap_mel2048_wave_kernel(ApMelWaveParams P) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    ap_float2 *X = reinterpret_cast<ap_float2 *>(ap_smem) + wave * APW_X_COMPLEX;
    ap_float2 *TW2 = reinterpret_cast<ap_float2 *>(ap_smem + P.off_tw2);     // [4][17]
    float *PP = reinterpret_cast<float *>(ap_smem + P.off_pp);               // [4][APW_PP_STRIDE]
    ap_float4 *WQ = reinterpret_cast<ap_float4 *>(ap_smem + P.off_wq);       // [n_quads]
    ap_int4 *PART = reinterpret_cast<ap_int4 *>(ap_smem + P.off_parts);      // [n_parts]
    float *PARTIAL = reinterpret_cast<float *>(ap_smem + P.off_partial);     // [n_parts][4]
    float *MACC = reinterpret_cast<float *>(ap_smem + P.off_macc);           // [M][16]
    const int M = P.n_mels;

    // ---------------- workgroup tables (once) ---------------------------------------
    for (int i = tid; i < P.n_quads; i += 256)
        WQ[i] = reinterpret_cast<const ap_float4 *>(P.quads)[i];
    for (int i = tid; i < P.n_parts; i += 256)
        PART[i] = reinterpret_cast<const ap_int4 *>(P.parts)[i];
    if (tid < 64) TW2[(tid >> 4) * 17 + (tid & 15)] = P.tw[32 * (tid >> 4) * (tid & 15)];   // W_64^(a*c)
    // ---------------- per-lane constants (registers) --------------------------------
    float win[32];
    ap_float2 tw1[16], tws[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        win[2 * j] = P.window[2 * (lane + 64 * j)];
        win[2 * j + 1] = P.window[2 * (lane + 64 * j) + 1];
        tw1[j] = P.tw[2 * lane * j];                 // W_1024^(lane*j)
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) tws[r] = P.tw[lane + 64 * r];    // W_2048^k, k = lane + 64 r
    // rows this thread sums in the gather pass: (row, frame) = (v >> 2, v & 3), v = tid + 256 i
    const int qa = lane & 3;                         // position in the quad
    const float s1 = qa < 2 ? 1.0f : -1.0f;          // radix-4 stage-1 sign
    const float s2 = (qa & 1) ? -1.0f : 1.0f;        // radix-4 stage-2 sign
    const bool rot = qa == 3;                        // lane 3 multiplies by -i between stages
    const int qd = ((qa & 1) << 1) | (qa >> 1);      // output digit held by this lane
    const int k1p = lane >> 2;
    const ap_float2 *tw2row = TW2 + qa * 17;
    float *pp = PP + wave * APW_PP_STRIDE;
    __syncthreads();

    for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        const int64_t b = tile / P.tiles_per_clip;
        const int64_t t0 = (tile - b * P.tiles_per_clip) * APW_G;
        const int Gt = (int)((P.T - t0) < APW_G ? (P.T - t0) : APW_G);
        const float *yb = P.y + b * P.L;

        ap_float2 raw[16];
        auto load_frame = [&](int g) {
            const int64_t base = (t0 + g) * (int64_t)P.hop - P.pad;
            if (base >= 0 && base + 2 * APW_NC <= P.L) {
                const float *src = yb + base;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    raw[j] = ap_mk(src[2 * lane + 128 * j], src[2 * lane + 128 * j + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int64_t p = base + 2 * (lane + 64 * j);
                    raw[j] = ap_mk(ap_load_padded(yb, P.L, p, P.pad_mode),
                                   ap_load_padded(yb, P.L, p + 1, P.pad_mode));
                }
            }
        };
        if (wave < Gt) load_frame(wave);

        for (int round = 0; round < APW_G / APW_WAVES; ++round) {
            const int g = round * APW_WAVES + wave;          // this wave's frame in the tile
            if (g < Gt) {
                ap_float2 v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = ap_mk(raw[j].x * win[2 * j], raw[j].y * win[2 * j + 1]);
                if (g + APW_WAVES < Gt) load_frame(g + APW_WAVES);   // in flight during this frame
                // ---- pass 1: radix-16 over j, twiddle W_1024^(lane*k1) -----------------
                ApButterfly<16>::run(v);
#pragma unroll
                for (int k = 1; k < 16; ++k) v[k] = ap_mul_fw(v[k], tw1[k]);
                // ---- transpose #1: (n0 = a + 4b, k1) -> lane (k1, a), register b -------
                {
                    const int a = lane & 3, bq = lane >> 2;
#pragma unroll
                    for (int k = 0; k < 16; ++k) X[(k * 4 + a) * APW_ROW + bq] = v[k];
                }
                AP_WAVE_SYNC();
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = X[lane * APW_ROW + i];
                AP_WAVE_SYNC();
                // ---- pass 2: radix-16 over b, twiddle W_64^(a*c) from the LDS table ------
                ApButterfly<16>::run(v);
#pragma unroll
                for (int c = 1; c < 16; ++c) v[c] = ap_mul_fw(v[c], tw2row[c]);
                // ---- pass 3: radix-4 across the quad (DIF), outputs in bit-reversed lanes -
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    float tx = ap_quad_xor2(v[c].x) + s1 * v[c].x;
                    float ty = ap_quad_xor2(v[c].y) + s1 * v[c].y;
                    const float rx = rot ? ty : tx;          // * (-i) on lane 3
                    const float ry = rot ? -tx : ty;
                    v[c].x = ap_quad_xor1(rx) + s2 * rx;
                    v[c].y = ap_quad_xor1(ry) + s2 * ry;
                }
                // ---- transpose #2: natural order Z[k], k = k1 + 16 c + 256 d ------------
#pragma unroll
                for (int c = 0; c < 16; ++c) X[apw_zidx(k1p + 16 * c + 256 * qd)] = v[c];
                AP_WAVE_SYNC();
                // ---- paired real split: bins k = lane + 64 r and 1024 - k ----------------
                //   a2 = Z[k] + conj Z[1024-k], d2 = Z[k] - conj Z[1024-k], u = W^k d2
                //   2X[k] = (a2.x + u.y, a2.y - u.x),  2X[1024-k] = (a2.x - u.y, -a2.y - u.x)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int k = lane + 64 * r;
                    const ap_float2 zk = X[apw_zidx(k)];
                    const ap_float2 zm = X[apw_zidx((APW_NC - k) & (APW_NC - 1))];
                    const float ax = zk.x + zm.x, ay = zk.y - zm.y;
                    const float dx = zk.x - zm.x, dy = zk.y + zm.y;
                    const float ux = tws[r].x * dx + tws[r].y * dy;
                    const float uy = tws[r].x * dy - tws[r].y * dx;
                    pp[k] = apw_pow2x<PMODE>(ax + uy, ay - ux, P.power);
                    pp[APW_NC - k] = apw_pow2x<PMODE>(ax - uy, -ay - ux, P.power);
                }
                if (lane == 0) {                                   // the unpaired bin 512
                    const ap_float2 zh = X[apw_zidx(APW_NC / 2)];
                    pp[APW_NC / 2] = apw_pow2x<PMODE>(2.0f * zh.x, 2.0f * zh.y, P.power);
                }
            }
            __syncthreads();
            // ---- mel contraction of the 4 planes of this round -------------------------
            for (int u = tid; u < 4 * P.n_parts; u += 256) {
                const ap_int4 pd = PART[u >> 2];                   // row, g0, ng, q0
                const int f = u & 3;
                const ap_float4 *pq = reinterpret_cast<const ap_float4 *>(PP + f * APW_PP_STRIDE) + pd.y;
                const ap_float4 *wq = WQ + pd.w;
                float acc = 0.0f;
                for (int i = 0; i < pd.z; ++i) {
                    const ap_float4 w = wq[i], p = pq[i];
                    acc = fmaf(w.x, p.x, acc);
                    acc = fmaf(w.y, p.y, acc);
                    acc = fmaf(w.z, p.z, acc);
                    acc = fmaf(w.w, p.w, acc);
                }
                PARTIAL[u] = acc;
            }
            __syncthreads();
            for (int vv = tid; vv < 4 * M; vv += 256) {
                const int row = vv >> 2, f = vv & 3;
                const int32_t *rp = P.rowparts + row * APW_RP;
                float sum = 0.0f;
                for (int j = 0; j < APW_RP; ++j) {
                    const int pid = rp[j];
                    if (pid < 0) break;
                    sum += PARTIAL[pid * 4 + f];
                }
                MACC[row * APW_G + round * APW_WAVES + f] = sum;
            }
        }
        __syncthreads();
        // ---- store the tile: 16 consecutive lanes write one 64-byte row segment -----
        float *ob = P.out + b * (int64_t)M * P.T + t0;
        for (int e = tid; e < M * APW_G; e += 256) {
            const int m = e >> 4, g = e & 15;
            if (g < Gt) ob[(int64_t)m * P.T + g] = MACC[e];
        }
    }
}
