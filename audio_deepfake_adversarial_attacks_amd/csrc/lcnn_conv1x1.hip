// lcnn_conv1x1.hip — LCNN's 1x1 "network-in-network" blocks fused on gfx950:
//     Conv2d(Cin, 2C, (1, 1)) -> MaxFeatureMap2D [-> BatchNorm2d(eval, affine=False)]
//     (src/models/lcnn.py:125-127, 132-134, 139-141, 146-148)
// forward and input-backward (C ABI: include/advstep_lcnn.h).
//
// A 1x1 convolution over NCHW is a skinny GEMM per pixel column — out (2C x P) = W (2C x Cin) . X (Cin x P) with
// K = Cin = 32..64 — and its output is consumed only by the max-feature-map.  Run separately (MIOpen GEMM + ATen add +
// max) the 2C-channel tensor is written and read twice (L3 at B = 128: 265 MB each way).  Here it is ONE kernel on the
// matrix cores: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, exact: the result is the k-ordered fmaf chain), the
// "MFMA where the work is a true dense contraction" part of the design.
//   * a wave owns 32 pixels: its B fragments (X[ci][px], one float per lane per K-step) are loaded straight from global
//     memory — 128 B contiguous per half wave — and live in Cin/2 registers;
//   * W is staged once per workgroup in LDS with an odd row pitch (conflict-free fragment reads);
//   * rows of the two channel halves land in the SAME (lane, register) slot of two accumulators, so the max-feature-map
//     is a register-to-register max; bias, the eval BatchNorm and the selection bit are the epilogue;
//   * selection bits (round 6): a lane keeps the bits of ITS OWN channels - bit 16 t + r for accumulator register r of
//     channel tile t - and stores them once per pixel tile: [n][pixel tile][pixel % 32][lane half] uint16 (C <= 32) or
//     uint32, 128 / 256 contiguous bytes per wave.  Rounds 1-5 wrote one ballot word per channel and 32-pixel tile
//     ([n][c][p / 32]): 16 four-byte stores per wave to 32 different rows, each a partial line that the 8 XCDs' L2s
//     wrote back separately - 10 us of the first block's 67 (profiles/r06_conv1x1_experiments.txt); the backward read
//     them with one broadcast load per k-step and now reads ONE word per lane;
//   * backward is the transposed GEMM over K = 2C with the max-feature-map backward applied while the B fragment is
//     formed (the gradient goes to the selected half's weight row), so the 2C-channel gradient never exists either.
// HBM traffic: Cin*4 B in, C*4 B + 1 bit out per pixel.  Accumulator map (32x32 tile, 16 registers): col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_lcnn.h"

namespace {

constexpr int kBlock = 256;          // 4 waves
constexpr int kPixPerBlock = 128;    // 32 pixels per wave

typedef float f32x16 __attribute__((ext_vector_type(16)));

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

__device__ __forceinline__ bool mfm_takes_b(float a, float b) { return !(a != a) && !(a >= b); }
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Selection format.  Channel c of a pixel lives in the forward lane half h = (c >> 2) & 1 of that pixel, bit
// 16 (c >> 5) + (c & 3) + 4 ((c & 31) >> 3) of that lane's mask (= 16 t + r for accumulator register r of tile t; in MODE 2's
// shared last tile r < 8).  Masks are uint16 for C <= 32 and uint32 above; the two halves of a pixel are adjacent.
// The backward's k-step s feeds channel c = 2 s + lk: h = (s >> 1) & 1 and the bit is sel_bit(s) + lk - all but lk known per step.
__host__ __device__ constexpr int sel_bit(int s) { return 16 * (s >> 4) + 2 * (s & 1) + 4 * ((s >> 2) & 3); }
__host__ __device__ constexpr int sel_half(int s) { return (s >> 1) & 1; }
inline int64_t sel_mask_bytes(int64_t C) { return C <= 32 ? 2 : 4; }

// grid (ceil(P / 128), N).  TILES = ceil(C / 32) channel tiles per half.
// LDS: w_s[2 * TILES * 32][CIN + 1]; rows [0, 32 TILES) = first half (zero beyond C), rows [32 TILES, 64 TILES) = second.
// MODE 1: C == 32 TILES (no accumulator row beyond C): the row-liveness branches of the epilogue compile away.
// MODE 2 (round 4): C == 32 TILES - 16 (LCNN's 48-channel block): the 16 leftover rows of BOTH halves share the last 32-row
// tile - rows 0..15 = channels 32 (TILES - 1) + i of the first half, rows 16..31 the same channels of the second half - so a
// max-feature-map pair is (acc[r], acc[r + 8]) of one lane, and the block issues 3 matrix instructions per k-step, not 4
// (the 64-row padding of 48 channels was a third of the kernel's matrix time).  Same k order per output: bit-identical.
template <int CIN, int TILES, int MODE>
__global__ __launch_bounds__(kBlock) void conv1x1_mfm_forward_kernel(const float *__restrict__ x,
                                                                     const float *__restrict__ weight,
                                                                     const float *__restrict__ bias,
                                                                     const float *__restrict__ bn_mean,
                                                                     const float *__restrict__ bn_invstd,
                                                                     float *__restrict__ y, uint8_t *__restrict__ sel,
                                                                     int C, int64_t P, int64_t PW) {
    extern __shared__ __attribute__((aligned(16))) float w_s[];
    constexpr int CP = TILES * 32, PITCH = CIN + 1;
    constexpr uint32_t SELB = TILES == 1 ? 2u : 4u;        // bytes of a lane's mask (the host picks TILES = 1 iff C <= 32)
    float *par_s = w_s + 2 * CP * PITCH;  // [4][CP]: bias of half a, bias of half b, BN mean, BN invstd
    // weights -> LDS in batches of kWBatch loads per thread, all issued before the first is used (a load -> LDS write
    // loop pays one L2 latency per iteration: 24-32 iterations cost more than the tile's matrix instructions)
    constexpr int kWLoads = 2 * CP * CIN / kBlock;
    constexpr int kWBatch = kWLoads % 8 == 0 ? 8 : 12;
    static_assert((2 * CP * CIN) % kBlock == 0 && kWLoads % kWBatch == 0, "weight staging batches");
#pragma unroll
    for (int j0 = 0; j0 < kWLoads; j0 += kWBatch) {
        float wv[kWBatch];
#pragma unroll
        for (int j = 0; j < kWBatch; ++j) {
            const int i = threadIdx.x + (j0 + j) * kBlock, row = i / CIN, ci = i - row * CIN;
            const int half = row / CP, c = row - half * CP;
            wv[j] = weight[(half * C + (c < C ? c : 0)) * CIN + ci];
        }
#pragma unroll
        for (int j = 0; j < kWBatch; ++j) {
            const int i = threadIdx.x + (j0 + j) * kBlock, row = i / CIN, ci = i - row * CIN;
            const int c = row % CP;
            w_s[row * PITCH + ci] = c < C ? wv[j] : 0.0f;
        }
    }
    for (int c = threadIdx.x; c < CP; c += kBlock) {
        const bool live = c < C;
        par_s[c] = (bias && live) ? bias[c] : 0.0f;
        par_s[CP + c] = (bias && live) ? bias[c + C] : 0.0f;
        par_s[2 * CP + c] = (bn_mean && live) ? bn_mean[c] : 0.0f;
        par_s[3 * CP + c] = (bn_mean && live) ? bn_invstd[c] : 1.0f;
    }
    __syncthreads();

    const int64_t n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    // one 32-pixel tile per wave (walking several tiles per wave with register double buffering measured 1.5-1.8x
    // slower: the live B fragments of two tiles push the kernel down to 1-2 waves per SIMD)
    const int64_t wave_p0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (wave_p0 >= P) return;  // wave-uniform: no pixel of this wave exists
    const int64_t p = wave_p0 + li;
    const bool valid = p < P;
    // Addressing is 32-bit and mostly scalar: one buffer descriptor per tensor of this sample, a per-lane byte offset
    // (row 4 lk or lk, pixel p — fixed for the whole kernel) and a wave-uniform row offset in an SGPR.  An out-of-range
    // pixel gets an out-of-range lane offset (loads return 0, stores are dropped).  (64-bit pointer arithmetic per access cost ~300 of this kernel's ~570 VALU
    // instructions, and VALU time adds to matrix time on this part: DESIGN.md section 4f.)
    const uint32_t Pb = (uint32_t)P * 4u;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + n * CIN * P), 0, (int)(CIN * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + n * (int64_t)C * P, 0, (int)(C * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr =
        __builtin_amdgcn_make_buffer_rsrc(sel + n * PW * (int64_t)(64u * SELB), 0, (int)((uint32_t)PW * 64u * SELB), 0x00020000);
    constexpr uint32_t kOut = 0x80000000u;
    const uint32_t pb = (uint32_t)p * 4u;
    const uint32_t x_off = valid ? (uint32_t)lk * Pb + pb : kOut;                          // row lk, pixel p
    const uint32_t y_off = valid ? 4u * (uint32_t)lk * Pb + pb : kOut;                     // row 4 lk, pixel p
    const uint32_t s_off = valid ? (uint32_t)(2 * li + lk) * SELB : kOut;                  // this lane's mask inside the wave's tile
    const uint32_t s_tile = (uint32_t)(wave_p0 >> 5) * 64u * SELB;
    uint32_t mask = 0;                                                                     // bit 16 t + r: the second half won
    float xb[CIN / 2];
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s)
        xb[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, x_off, (uint32_t)(2 * s) * Pb, 0));

    constexpr bool FULL = MODE != 0;                       // rows of the two-accumulator tiles are all live
    constexpr int kPairTiles = MODE == 2 ? TILES - 1 : TILES;
#pragma unroll
    for (int t = 0; t < kPairTiles; ++t) {
        f32x16 acc_a = {0}, acc_b = {0};
        const float *wa = w_s + (t * 32 + li) * PITCH + lk;
        const float *wb = wa + CP * PITCH;
#pragma unroll
        for (int s = 0; s < CIN / 2; ++s) {
            acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * s], xb[s], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[2 * s], xb[s], acc_b, 0, 0, 0);
        }
        // branch-free epilogue: per-channel parameters come from LDS (identity values where absent)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c_lo = t * 32 + (r & 3) + 8 * (r >> 2);        // lanes 0-31: channel c_lo, lanes 32-63: c_lo + 4
            const int c = c_lo + 4 * lk;
            const float va = acc_a[r] + par_s[c], vb = acc_b[r] + par_s[CP + c];
            const bool tb = mfm_takes_b(va, vb);
            mask |= tb ? (1u << (16 * t + r)) : 0u;      // (rows >= C: zero weights and parameters, the bit stays 0)
            const float v = ((tb ? vb : va) - par_s[2 * CP + c]) * par_s[3 * CP + c];
            // the descriptor's range check covers the lane offset only, NOT the scalar row offset: rows >= C must not be
            // stored.  Wave-uniform cases first (C a multiple of 8 never takes the per-lane one).
            if (FULL || c_lo + 4 < C) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, y_off, (uint32_t)c_lo * Pb, 0);
            } else if (c_lo < C) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, lk ? kOut : y_off,
                                                      (uint32_t)c_lo * Pb, 0);
            }
        }
    }
    if constexpr (MODE == 2) {
        constexpr int t = TILES - 1;
        f32x16 acc = {0};
        const float *wm = w_s + (li < 16 ? t * 32 + li : CP + t * 32 + (li - 16)) * PITCH + lk;
#pragma unroll
        for (int s = 0; s < CIN / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wm[2 * s], xb[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int c_lo = t * 32 + (r & 3) + 8 * (r >> 2);        // lanes 0-31: channel c_lo, lanes 32-63: c_lo + 4
            const int c = c_lo + 4 * lk;
            const float va = acc[r] + par_s[c], vb = acc[r + 8] + par_s[CP + c];
            const bool tb = mfm_takes_b(va, vb);
            mask |= tb ? (1u << (16 * t + r)) : 0u;
            const float v = ((tb ? vb : va) - par_s[2 * CP + c]) * par_s[3 * CP + c];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, y_off, (uint32_t)c_lo * Pb, 0);
        }
    }
    // one selection store per wave: 64 lanes x SELB contiguous bytes
    if constexpr (SELB == 2) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)mask, sr, s_off, s_tile, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(mask, sr, s_off, s_tile, 0);
}

// grid (ceil(P / 128), N).  MT = ceil(CIN / 32) output tiles (input channels).  LDS: w_s[2C][32 MT + 1], columns >= CIN zero.
// STEPS = C / 2 when known at compile time (LCNN: 16, 24, 32): every gradient / selection load of the wave is issued
// before the first MFMA; STEPS = 0 is the generic runtime loop.
template <int CIN, int MT, int STEPS>
__global__ __launch_bounds__(kBlock) void conv1x1_mfm_backward_kernel(const float *__restrict__ gy,
                                                                      const uint8_t *__restrict__ sel,
                                                                      const float *__restrict__ weight,
                                                                      const float *__restrict__ gscale,
                                                                      float *__restrict__ gx, int C, int64_t P,
                                                                      int64_t PW) {
    extern __shared__ __attribute__((aligned(16))) float w_s[];
    constexpr int CW = MT * 32, PITCH = CW + 1;
    float *gs_s = w_s + 2 * C * PITCH;  // [C]: the following BatchNorm's invstd (1 where absent)
    // weights -> LDS, several loads per thread in flight (see the forward kernel); with C known at compile time
    // (STEPS > 0: C = 2 STEPS) the index arithmetic folds away
    if constexpr (STEPS > 0) {
        constexpr int kRows = 4 * STEPS, kWLoads = kRows * CW / kBlock, kWBatch = kWLoads % 8 == 0 ? 8 : 12;
        static_assert((kRows * CW) % kBlock == 0 && kWLoads % kWBatch == 0, "weight staging batches");
#pragma unroll
        for (int j0 = 0; j0 < kWLoads; j0 += kWBatch) {
            float wv[kWBatch];
#pragma unroll
            for (int j = 0; j < kWBatch; ++j) {
                const int i = threadIdx.x + (j0 + j) * kBlock, row = i / CW, ci = i - row * CW;
                wv[j] = weight[row * CIN + (ci < CIN ? ci : 0)];
            }
#pragma unroll
            for (int j = 0; j < kWBatch; ++j) {
                const int i = threadIdx.x + (j0 + j) * kBlock, row = i / CW, ci = i - row * CW;
                w_s[row * PITCH + ci] = ci < CIN ? wv[j] : 0.0f;
            }
        }
    } else {
        for (int i = threadIdx.x; i < 2 * C * CW; i += kBlock) {
            const int row = i / CW, ci = i - row * CW;
            w_s[row * PITCH + ci] = ci < CIN ? weight[(int64_t)row * CIN + ci] : 0.0f;
        }
    }
    for (int c = threadIdx.x; c < C; c += kBlock) gs_s[c] = gscale ? gscale[c] : 1.0f;
    __syncthreads();

    const int64_t n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const int64_t p = (int64_t)blockIdx.x * kPixPerBlock + wave * 32 + li;
    const bool valid = p < P;
    if ((int64_t)blockIdx.x * kPixPerBlock + wave * 32 >= P) return;

    // 32-bit, mostly scalar addressing as in the forward kernel
    const uint32_t Pb = (uint32_t)P * 4u;
    const bool wide = C > 32;                    // uint32 masks (the forward's TILES = 2), else uint16
    const uint32_t pairb = wide ? 8u : 4u;       // bytes of a pixel's two masks
    const __amdgpu_buffer_rsrc_t gr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gy + n * (int64_t)C * P), 0, (int)(C * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(sel + n * PW * (int64_t)(32u * pairb)), 0, (int)((uint32_t)PW * 32u * pairb), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(gx + n * CIN * P, 0, (int)(CIN * Pb), 0x00020000);
    constexpr uint32_t kOut = 0x80000000u;
    const uint32_t pb = (uint32_t)p * 4u;
    const uint32_t g_off = valid ? (uint32_t)lk * Pb + pb : kOut;                              // row lk, pixel p
    const uint32_t x_off = valid ? 4u * (uint32_t)lk * Pb + pb : kOut;                         // row 4 lk, pixel p
    // the pixel's two masks (forward lane halves 0 and 1) in ONE load per lane; m[h] >> lk puts the bit of channel 2 s + lk
    // at sel_bit(s).  (Rounds 1-5: one broadcast word per k-step.)
    const uint32_t s_off = valid ? (uint32_t)li * pairb : kOut;
    const uint32_t s_tile = (uint32_t)(((int64_t)blockIdx.x * kPixPerBlock + wave * 32) >> 5) * 32u * pairb;
    uint32_t m[2];
    if (wide) {
        m[0] = __builtin_amdgcn_raw_buffer_load_b32(sr, s_off, s_tile, 0) >> lk;
        m[1] = __builtin_amdgcn_raw_buffer_load_b32(sr, s_off + (valid ? 4u : 0u), s_tile, 0) >> lk;
    } else {
        const uint32_t both = __builtin_amdgcn_raw_buffer_load_b32(sr, s_off, s_tile, 0);
        m[0] = (both & 0xffffu) >> lk;
        m[1] = (both >> 16) >> lk;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x16){0};
    // K runs over the 2C rows of W^T; step s feeds channel pair c = 2 s + lk to BOTH halves: the gradient reaches the
    // selected half only (max-feature-map backward), the other B fragment is zero
    if constexpr (STEPS > 0) {
        float g[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            g[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, g_off, (uint32_t)(2 * s) * Pb, 0));
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = 2 * s + lk;
            const float gs = g[s] * gs_s[c];
            const bool tb = (m[sel_half(s)] >> sel_bit(s)) & 1u;   // the second half won for channel c at this lane's pixel
            const float ga = tb ? 0.0f : gs, gb = tb ? gs : 0.0f;
            const float *wa = w_s + c * PITCH + li;
            const float *wb = wa + C * PITCH;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[32 * m], ga, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[32 * m], gb, acc[m], 0, 0, 0);
            }
        }
    } else {
    const int steps = (C + 1) >> 1;
    for (int s = 0; s < steps; ++s) {
        const int c = 2 * s + lk;
        const bool live = c < C;
        const float g = live ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, g_off, (uint32_t)(2 * s) * Pb, 0)) *
                                   gs_s[c]
                             : 0.0f;
        const bool tb = live && ((m[sel_half(s)] >> sel_bit(s)) & 1u);
        const float ga = tb ? 0.0f : g, gb = tb ? g : 0.0f;
        const float *wa = w_s + (live ? c : 0) * PITCH + li;
        const float *wb = w_s + ((live ? c : 0) + C) * PITCH + li;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(live ? wa[32 * m] : 0.0f, ga, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(live ? wb[32 * m] : 0.0f, gb, acc[m], 0, 0, 0);
        }
    }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // lanes 32-63 hold row ci_lo + 4.  Rows >= CIN must not be stored (the range check does not see the scalar
            // row offset); CIN is a multiple of 8, so a row pair is live or dead as a whole — decided at compile time
            const int ci_lo = m * 32 + (r & 3) + 8 * (r >> 2);
            static_assert(CIN % 8 == 0, "row pairs (ci, ci + 4) must be live or dead together");
            const float gv = acc[m][r];   // (bit-casting the vector element in place stores element 0 sixteen times: hipcc 7.2)
            if (ci_lo < CIN)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gv), xr, x_off, (uint32_t)ci_lo * Pb, 0);
        }
    }
}

constexpr int64_t kMaxGridY = 65535;

template <typename K>
void opt_in_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

template <int CIN, int TILES>
void launch_fwd(const float *x, const float *w, const float *b, const float *bn_mean, const float *bn_invstd, float *y,
                uint8_t *sel, int64_t N, int64_t C, int64_t P, hipStream_t st) {
    const size_t lds = (size_t)(2 * TILES * 32 * (CIN + 1) + 4 * TILES * 32) * sizeof(float);
    const dim3 grid((unsigned)ceil_div(P, kPixPerBlock), (unsigned)N);
    if (C == TILES * 32) {
        opt_in_lds(conv1x1_mfm_forward_kernel<CIN, TILES, 1>, lds);
        hipLaunchKernelGGL((conv1x1_mfm_forward_kernel<CIN, TILES, 1>), grid, dim3(kBlock), lds, st, x, w, b, bn_mean,
                           bn_invstd, y, sel, (int)C, P, ceil_div(P, 32));
    } else if (C == TILES * 32 - 16) {
        opt_in_lds(conv1x1_mfm_forward_kernel<CIN, TILES, 2>, lds);
        hipLaunchKernelGGL((conv1x1_mfm_forward_kernel<CIN, TILES, 2>), grid, dim3(kBlock), lds, st, x, w, b, bn_mean,
                           bn_invstd, y, sel, (int)C, P, ceil_div(P, 32));
    } else {
        opt_in_lds(conv1x1_mfm_forward_kernel<CIN, TILES, 0>, lds);
        hipLaunchKernelGGL((conv1x1_mfm_forward_kernel<CIN, TILES, 0>), grid, dim3(kBlock), lds, st, x, w, b, bn_mean,
                           bn_invstd, y, sel, (int)C, P, ceil_div(P, 32));
    }
}

template <int CIN, int MT, int STEPS>
void launch_bwd_steps(const float *gy, const uint8_t *sel, const float *w, const float *gscale, float *gx, int64_t N,
                      int64_t C, int64_t P, hipStream_t st) {
    const size_t lds = (size_t)(2 * C * (MT * 32 + 1) + C) * sizeof(float);
    opt_in_lds(conv1x1_mfm_backward_kernel<CIN, MT, STEPS>, lds);
    const dim3 grid((unsigned)ceil_div(P, kPixPerBlock), (unsigned)N);
    hipLaunchKernelGGL((conv1x1_mfm_backward_kernel<CIN, MT, STEPS>), grid, dim3(kBlock), lds, st, gy, sel, w, gscale, gx,
                       (int)C, P, ceil_div(P, 32));
}

// LCNN's blocks have C == Cin (the conv doubles the channels, the max-feature-map halves them): that case gets the
// fully unrolled kernel, anything else the runtime loop
template <int CIN, int MT>
void launch_bwd(const float *gy, const uint8_t *sel, const float *w, const float *gscale, float *gx, int64_t N, int64_t C,
                int64_t P, hipStream_t st) {
    if (C == CIN)
        launch_bwd_steps<CIN, MT, CIN / 2>(gy, sel, w, gscale, gx, N, C, P, st);
    else
        launch_bwd_steps<CIN, MT, 0>(gy, sel, w, gscale, gx, N, C, P, st);
}

}  // namespace

#define C11_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

int advstep_conv1x1_mfm_supported(int64_t Cin) { return Cin == 32 || Cin == 48 || Cin == 64; }

size_t advstep_conv1x1_mfm_sel_bytes(int64_t N, int64_t C, int64_t P) {
    if (N <= 0 || C <= 0 || P <= 0) return 0;
    return (size_t)N * (size_t)ceil_div(P, 32) * 64u * (size_t)sel_mask_bytes(C);   // a mask per lane of every 32-pixel tile
}

int advstep_conv1x1_mfm_forward_f32(const float *x, const float *weight, const float *bias, const float *bn_mean,
                                    const float *bn_invstd, float *y, void *sel, int64_t N, int64_t Cin, int64_t C,
                                    int64_t P, advstep_stream_t stream) {
    C11_REQUIRE(N >= 0 && C >= 0 && P >= 0 && advstep_conv1x1_mfm_supported(Cin));
    if (N == 0 || C == 0 || P == 0) return ADVSTEP_OK;
    C11_REQUIRE(x && weight && y && sel && N <= kMaxGridY && C <= 64 && ((reinterpret_cast<uintptr_t>(sel) & 3u) == 0));
    C11_REQUIRE((bn_mean == nullptr) == (bn_invstd == nullptr));
    auto *s32 = static_cast<uint8_t *>(sel);
    hipStream_t st = as_stream(stream);
    const bool two = C > 32;
    switch (Cin) {
        case 32:
            two ? launch_fwd<32, 2>(x, weight, bias, bn_mean, bn_invstd, y, s32, N, C, P, st)
                : launch_fwd<32, 1>(x, weight, bias, bn_mean, bn_invstd, y, s32, N, C, P, st);
            break;
        case 48:
            two ? launch_fwd<48, 2>(x, weight, bias, bn_mean, bn_invstd, y, s32, N, C, P, st)
                : launch_fwd<48, 1>(x, weight, bias, bn_mean, bn_invstd, y, s32, N, C, P, st);
            break;
        default:
            two ? launch_fwd<64, 2>(x, weight, bias, bn_mean, bn_invstd, y, s32, N, C, P, st)
                : launch_fwd<64, 1>(x, weight, bias, bn_mean, bn_invstd, y, s32, N, C, P, st);
            break;
    }
    return status_after_launch();
}

int advstep_conv1x1_mfm_backward_f32(const float *gy, const void *sel, const float *weight, const float *gscale, float *gx,
                                     int64_t N, int64_t Cin, int64_t C, int64_t P, advstep_stream_t stream) {
    C11_REQUIRE(N >= 0 && C >= 0 && P >= 0 && advstep_conv1x1_mfm_supported(Cin));
    if (N == 0 || P == 0) return ADVSTEP_OK;
    C11_REQUIRE(gx && N <= kMaxGridY && C <= 64);
    hipStream_t st = as_stream(stream);
    if (C == 0)
        return hipMemsetAsync(gx, 0, (size_t)N * Cin * P * sizeof(float), st) == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH;
    C11_REQUIRE(gy && sel && weight);
    C11_REQUIRE((reinterpret_cast<uintptr_t>(sel) & 3u) == 0);
    auto *s32 = static_cast<const uint8_t *>(sel);
    switch (Cin) {
        case 32: launch_bwd<32, 1>(gy, s32, weight, gscale, gx, N, C, P, st); break;
        case 48: launch_bwd<48, 2>(gy, s32, weight, gscale, gx, N, C, P, st); break;
        default: launch_bwd<64, 2>(gy, s32, weight, gscale, gx, N, C, P, st); break;
    }
    return status_after_launch();
}

}  // extern "C"
