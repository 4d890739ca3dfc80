// lcnn_conv0.hip — LCNN's first block fused on gfx950:
//     Conv2d(1, 2C, (5, 5), padding 2) -> MaxFeatureMap2D -> MaxPool2d((2, 2), (2, 2))      (src/models/lcnn.py:121-123)
// forward and input-backward (C ABI: include/advstep_lcnn.h).
//
// Why: with one input channel the convolution is not GEMM-shaped (K = 25) — it is a 1.06 GB write of an
// activation ((B, 64, 404, 80) f32 at B = 128) whose only consumer is the max-feature-map + pool.  Fused, the
// conv output never exists: the forward reads the 16.5 MB spectrogram (L2 resident) and writes the pooled
// (B, 32, 202, 40) tensor + one selection byte per output; the backward gathers, for every 2x2 input patch,
// the <= 9 pooled cells whose winner can reach it, straight from gy + the selection bytes.
//   forward : thread = one pooled output position, 6x6 input window in registers, loop over the C channel
//             pairs with the 2 x 25 taps + 2 biases as wave-uniform (scalar) operands: 100 v_pk_fma_f32 per pair.
//   backward: (round 5, even widths up to 128) thread = one pooled CELL: its winner's 6x6 window of taps comes out of an LDS
//             table indexed by (channel, selection code), accumulates over the channels in registers, and the windows' 3x3
//             blocks meet in one exchange per tile - 9 LDS reads + 2 loads per cell and channel; LDS-bound (the useful
//             arithmetic is an eighth of the forward's).  Other shapes: thread = one 2x2 input patch gathering from its 9
//             pooled cells (rounds 1-4: 27 LDS reads + 18 loads per patch and channel).  No atomics either way.
// Forward: VALU-bound (13.2 GFLOP at B = 128), not HBM-bound; MFMA does not apply (K = 25, one input channel).
// Accumulation order is fixed (taps row-major, then channels), fma contraction allowed: deterministic, and within
// float rounding of any other convolution implementation (MIOpen's differs in the last bits as well).

#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "advstep_lcnn.h"

namespace {

constexpr int kBlock = 256;
constexpr int K = 5, KK = 25, PAD = 2;
typedef float f32x2 __attribute__((ext_vector_type(2)));

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

__device__ __forceinline__ bool mfm_takes_b(float a, float b) { return !(a != a) && !(a >= b); }

// same selection rule as lcnn_mfm.hip::pool_select
__device__ __forceinline__ float pool_select(float a00, float b00, float a01, float b01, float a10, float b10,
                                             float a11, float b11, int &code) {
    const bool t00 = mfm_takes_b(a00, b00), t01 = mfm_takes_b(a01, b01);
    const bool t10 = mfm_takes_b(a10, b10), t11 = mfm_takes_b(a11, b11);
    const float m00 = t00 ? b00 : a00, m01 = t01 ? b01 : a01, m10 = t10 ? b10 : a10, m11 = t11 ? b11 : a11;
    float best = -INFINITY;
    int pos = 0;
    bool tb = t00;
    if (m00 > best || m00 != m00) { best = m00; pos = 0; tb = t00; }
    if (m01 > best || m01 != m01) { best = m01; pos = 1; tb = t01; }
    if (m10 > best || m10 != m10) { best = m10; pos = 2; tb = t10; }
    if (m11 > best || m11 != m11) { best = m11; pos = 3; tb = t11; }
    code = ((int)tb << 2) | pos;
    return best;
}

// The same selection in 4 + 15 instead of ~33 vector instructions (round 3): gfx950's v_maximum3_f32 propagates NaN, so the
// maximum of the 8 candidates is the pooled value whenever no candidate is NaN, and the winner's code is the FIRST candidate
// in the reference's scan order (a00, b00, a01, b01, a10, b10, a11, b11: `a` keeps ties inside a position, the earlier position
// keeps ties between positions) that equals it.  A NaN among the candidates (best != best) takes the step-by-step rule above.
// (A tie between -0 and +0 returns +0 where the scan returns the first: convolution outputs, not bit-compared.)
__device__ __forceinline__ float pool_select_fast(float a00, float b00, float a01, float b01, float a10, float b10,
                                                  float a11, float b11, int &code) {
    const float best = __builtin_elementwise_maximum(
        __builtin_elementwise_maximum(__builtin_elementwise_maximum(a00, b00), __builtin_elementwise_maximum(a01, b01)),
        __builtin_elementwise_maximum(__builtin_elementwise_maximum(a10, b10), __builtin_elementwise_maximum(a11, b11)));
    if (best != best) return pool_select(a00, b00, a01, b01, a10, b10, a11, b11, code);
    int c = 7;
    c = a11 == best ? 3 : c;
    c = b10 == best ? 6 : c;
    c = a10 == best ? 2 : c;
    c = b01 == best ? 5 : c;
    c = a01 == best ? 1 : c;
    c = b00 == best ? 4 : c;
    c = a00 == best ? 0 : c;
    code = c;
    return best;
}

__global__ __launch_bounds__(kBlock) void conv5_mfm_pool2_forward_kernel(const float *__restrict__ x,
                                                                         const float *__restrict__ weight,
                                                                         const float *__restrict__ bias,
                                                                         float *__restrict__ y,
                                                                         uint8_t *__restrict__ idx, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t n = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int ho = p / Wo, wo = p - ho * Wo;
    const float *xn = x + n * (int64_t)H * W;

    // 6x6 input window around the 2x2 conv positions (zero padding)
    float win[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int h = 2 * ho - PAD + r;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int w = 2 * wo - PAD + q;
            win[r][q] = (h >= 0 && h < H && w >= 0 && w < W) ? xn[(int64_t)h * W + w] : 0.0f;
        }
    }

    // horizontally adjacent conv positions share a tap: (x00, x01) and (x10, x11) accumulate as float2 pairs, one
    // v_pk_fma_f32 per pair with the tap broadcast from a scalar register (100 packed FMAs per channel pair instead
    // of 200 scalar ones).  Window pairs are kept at even and at odd column offsets so every operand is an aligned
    // register pair.
    f32x2 pe[6][3], po[6][2];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int q = 0; q < 3; ++q) pe[r][q] = (f32x2){win[r][2 * q], win[r][2 * q + 1]};
#pragma unroll
        for (int q = 0; q < 2; ++q) po[r][q] = (f32x2){win[r][2 * q + 1], win[r][2 * q + 2]};
    }

    float *yn = y + n * (int64_t)C * Ho * Wo + p;
    uint8_t *in = idx + n * (int64_t)C * Ho * Wo + p;
    for (int c = 0; c < C; ++c) {
        const float *wa = weight + c * KK;        // wave-uniform addresses: scalar loads
        const float *wb = weight + (c + C) * KK;
        f32x2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f}, b0 = {0.0f, 0.0f}, b1 = {0.0f, 0.0f};  // rows 0 / 1 of the 2x2 block
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const float ua = wa[kh * K + kw], ub = wb[kh * K + kw];
                const f32x2 t0 = (kw & 1) ? po[kh][kw >> 1] : pe[kh][kw >> 1];
                const f32x2 t1 = (kw & 1) ? po[kh + 1][kw >> 1] : pe[kh + 1][kw >> 1];
                a0 = __builtin_elementwise_fma((f32x2){ua, ua}, t0, a0);
                a1 = __builtin_elementwise_fma((f32x2){ua, ua}, t1, a1);
                b0 = __builtin_elementwise_fma((f32x2){ub, ub}, t0, b0);
                b1 = __builtin_elementwise_fma((f32x2){ub, ub}, t1, b1);
            }
        }
        int code;
        // bias AFTER the taps, as the reference's convolution adds it (starting the accumulators at the bias was measured: it
        // re-routes 34 near-tie winners of 33 M against the MIOpen path on the LFCC input, whose sums are ~1e3 x the bias);
        // four packed adds
        const float ba = bias ? bias[c] : 0.0f, bb = bias ? bias[c + C] : 0.0f;
        a0 += (f32x2){ba, ba};
        a1 += (f32x2){ba, ba};
        b0 += (f32x2){bb, bb};
        b1 += (f32x2){bb, bb};
        const float v = pool_select_fast(a0.x, b0.x, a0.y, b0.y, a1.x, b1.x, a1.y, b1.y, code);
        yn[(int64_t)c * Ho * Wo] = v;
        in[(int64_t)c * Ho * Wo] = (uint8_t)code;
    }
}

// thread = one pooled-cell footprint (2x2 input patch); gathers from the 3x3 neighbouring pooled cells.
// Branch-free: the taps live in LDS as zero-bordered 7x7 tables (tap (kh, kw) at [kh + 1][kw + 1]), so a winner that
// cannot reach a patch pixel reads a 0 weight instead of taking a branch, and out-of-range neighbour cells read a
// clamped address with their gradient forced to 0.  The two channel halves are kHalfPitch words apart, which puts the
// 8 (half, dh, dw) variants of one lookup in 8 different LDS banks.
constexpr int kTab = 7, kTabWords = kTab * kTab;   // 49

__global__ __launch_bounds__(kBlock) void conv5_mfm_pool2_backward_kernel(const float *__restrict__ gy,
                                                                          const uint8_t *__restrict__ idx,
                                                                          const float *__restrict__ weight,
                                                                          float *__restrict__ gx, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [2][C][7][7] + 4 words between the halves, then lut[8]
    const int half_pitch = C * kTabWords + 4;
    for (int i = threadIdx.x; i < 2 * half_pitch; i += kBlock) wl[i] = 0.0f;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C * KK; i += kBlock) {
        const int ch = i / KK, tap = i - ch * KK, kh = tap / K, kw = tap - kh * K;
        const int half = ch >= C, c = ch - half * C;
        wl[half * half_pitch + c * kTabWords + (kh + 1) * kTab + (kw + 1)] = weight[i];
    }
    int *lut = reinterpret_cast<int *>(wl + 2 * half_pitch);   // byte offset of a winner's taps, by selection code
    if (threadIdx.x < 8) {
        const int code = threadIdx.x;
        lut[code] = 4 * (((code & 4) ? half_pitch : 0) - kTab * ((code >> 1) & 1) - (code & 1));
    }
    __syncthreads();

    const int Ho = H >> 1, Wo = W >> 1;
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;  // patches cover a trailing odd row / column too
    const int64_t n = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= Hp * Wp) return;
    const int hp = p / Wp, wp = p - hp * Wp;
    // gy and the selection bytes of this sample through buffer descriptors: a 32-bit lane offset per neighbour cell
    // (hoisted), the channel as a scalar offset — no per-access VALU address arithmetic (VALU time is what bounds this
    // kernel).  A neighbour outside the pooled grid gets an out-of-range offset: its gradient and code read as 0.
    const uint32_t plane = (uint32_t)(Ho * Wo);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(gy + n * (int64_t)C * plane), 0, (int)((uint32_t)C * plane * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(idx + n * (int64_t)C * plane), 0, (int)((uint32_t)C * plane), 0x00020000);
    uint32_t cell_b[9], cell_f[9];   // byte offsets into the selection bytes / the gradients; 0x80000000 = outside
#pragma unroll
    for (int dh = -1; dh <= 1; ++dh) {
#pragma unroll
        for (int dw = -1; dw <= 1; ++dw) {
            const int ho = hp + dh, wo = wp + dw;
            const bool ok = ho >= 0 && ho < Ho && wo >= 0 && wo < Wo;
            cell_b[(dh + 1) * 3 + dw + 1] = ok ? (uint32_t)(ho * Wo + wo) : 0x80000000u;
            cell_f[(dh + 1) * 3 + dw + 1] = ok ? (uint32_t)(ho * Wo + wo) * 4u : 0x80000000u;
        }
    }

    // (x00, x01) and (x10, x11) accumulate as float2 pairs: the two taps a winner sends to one patch row are adjacent
    // words of the bordered table (one ds_read2_b32 = an aligned register pair), the gradient is the broadcast operand.
    // Where the winner's taps start — which half's table, shifted by the winner's row / column inside its cell — is a
    // function of the 3 code bits: an 8-entry byte-offset table in LDS (one ds_read instead of six VALU instructions).
    f32x2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
    const char *lut_b = reinterpret_cast<const char *>(lut);
    // (Round 3, measured and not kept: requesting channel c + 1's nine (selection byte, gradient) pairs before channel c's are
    // consumed — two register sets, scheduling barriers — 150 -> 159 us; the offset table as five VALU instructions instead of
    // an LDS read: 161 us.  The loop is bound by its 27 LDS reads per channel, not by the latency of its 18 vector-memory loads.
    // Round 4: the (cell, code) windows expanded once per workgroup into float4 entries - 9 ds_read_b128 per channel instead of
    // 27 reads, bit-identical - 156 -> 184 us: 36 KB of LDS per workgroup leaves 16 waves per CU.)
    for (int c = 0; c < C; ++c) {
        const uint32_t gsoff = (uint32_t)c * plane * 4u, isoff = (uint32_t)c * plane;   // wave-uniform
        const char *wc = reinterpret_cast<const char *>(wl + c * kTabWords);
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                const int q = (dh + 1) * 3 + dw + 1;
                const uint32_t code = __builtin_amdgcn_raw_buffer_load_b8(ir, cell_b[q], isoff, 0);
                const float g = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, cell_f[q], gsoff, 0));
                // winner's conv position relative to the patch origin: (2 dh + ph, 2 dw + pw); patch pixel (i, j) sees
                // tap (i - rh + 2, j - rw + 2), stored at [i - rh + 3][j - rw + 3] of the bordered table: a constant
                // per neighbour plus lut[code]
                const int by_code = *reinterpret_cast<const int *>(lut_b + code * 4u);   // codes are 0..7 (forward kernel)
                const float *t0 = reinterpret_cast<const float *>(wc + ((3 - 2 * dh) * kTab + (3 - 2 * dw)) * 4 + by_code);
                const f32x2 gg = {g, g};
                acc0 = __builtin_elementwise_fma(gg, (f32x2){t0[0], t0[1]}, acc0);
                acc1 = __builtin_elementwise_fma(gg, (f32x2){t0[kTab], t0[kTab + 1]}, acc1);
            }
        }
    }
    const float acc00 = acc0.x, acc01 = acc0.y, acc10 = acc1.x, acc11 = acc1.y;
    float *xn = gx + n * (int64_t)H * W;
    const int h0 = 2 * hp, w0 = 2 * wp;
    xn[(int64_t)h0 * W + w0] = acc00;
    if (w0 + 1 < W) xn[(int64_t)h0 * W + w0 + 1] = acc01;
    if (h0 + 1 < H) {
        xn[(int64_t)(h0 + 1) * W + w0] = acc10;
        if (w0 + 1 < W) xn[(int64_t)(h0 + 1) * W + w0 + 1] = acc11;
    }
}

// ---- round 5: the same gradient, cell-centric -------------------------------------------------------------------------------
// The gather above reads, per 2x2 input patch and channel, the (gradient, code) pair of 9 pooled cells and 9 x 3 LDS words
// (27 LDS instructions, 18 vector-memory loads) for 36 multiply-adds: the LDS - one per compute unit - is what bounds it
// (profiles/r04_model_kernel_pmc_deep.txt).  Turned round: a thread owns a pooled CELL.  Its winner (channel half, position
// inside the 2x2 window) sends the 5x5 taps to a 6x6 window of input pixels around the cell; that window is a function of the
// channel and the 3-bit code alone, so all 8 of them sit expanded in LDS (36 floats each: the taps shifted by the winner's
// position, zeros around) and a cell reads its window with nine 16-byte loads at `code * 144`: per cell and channel 2
// vector-memory loads and 9 LDS reads for the same 36 multiply-adds, accumulated over all channels in 36 registers.  Only then
// do the 3x3 blocks of a window go to the patches they belong to: ONE exchange through LDS per tile (nine phases of one float4
// per cell), after which a thread holds the finished gradient of the 2x2 patch under its own cell.  A tile is the full plane
// width (no halo columns: cells beyond the plane do not exist) by band x CPT rows, of which the outer two are halo (computed
// twice).  Deterministic: fixed summation order, no atomics.
constexpr int kWinFloats = 36, kWinBytes = kWinFloats * 4;      // 144 B: the 8 windows of a channel start in 8 different
constexpr int kChanBytes = 8 * kWinBytes;                       // 16-byte bank groups (144 k mod 256 is a permutation)
constexpr int kCellsMaxWidth = 64;

template <int CPT>
__global__ __launch_bounds__(kBlock) void conv5_mfm_pool2_backward_cells_kernel(const float *__restrict__ gy,
                                                                                const uint8_t *__restrict__ idx,
                                                                                const float *__restrict__ weight,
                                                                                float *__restrict__ gx, int C, int H, int W,
                                                                                int TR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Ho = H >> 1, Wo = W >> 1, Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
    const int TRo = TR - 2, cells = TR * Wp;       // TR = (CPT * 256) / Wp tile rows; cell k * 256 + thread of the tile, row-major
    char *tab = smem;
    float4 *xch = reinterpret_cast<float4 *>(smem);        // [3][TR * Wp], over the tables once the channel loop is done
    {
        float4 *t4 = reinterpret_cast<float4 *>(tab);
        for (int i = threadIdx.x; i < C * kChanBytes / 16; i += kBlock) t4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __syncthreads();
        // one (channel, tap) per thread and trip, 32 lanes per channel (25 used): no integer division in the fill - at 25
        // trips of ~70 instructions the first version's fill cost as much as the channel loop - and the weights of 8 trips are
        // requested together (one L2 round trip per batch, not per trip)
        for (int i0 = threadIdx.x; i0 < 2 * C * 32; i0 += 8 * kBlock) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * kBlock, ch = i >> 5, tap = i & 31;
                w[u] = (tap < KK && ch < 2 * C) ? weight[ch * KK + tap] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * kBlock, ch = i >> 5, tap = i & 31;
                if (tap < KK && ch < 2 * C) {
                    const int kh = (tap * 13) >> 6, kw = tap - kh * K;          // tap / 5 for tap < 25
                    const int half = ch >= C, c = ch - half * C;
                    float *win = reinterpret_cast<float *>(tab + c * kChanBytes + (half << 2) * kWinBytes) + kh * 6 + kw;
                    // window[r][s] = tap (r - ph, s - pw) of the winner's filter: four copies, shifted by (ph, pw)
                    win[0] = w[u];
                    win[kWinFloats + 1] = w[u];
                    win[2 * kWinFloats + 6] = w[u];
                    win[3 * kWinFloats + 7] = w[u];
                }
            }
        }
        __syncthreads();
    }
    const uint32_t plane = (uint32_t)(Ho * Wo);
    const int64_t n = blockIdx.y;
    const int t0 = blockIdx.x * TRo;    // first output patch row of the tile
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(gy + n * (int64_t)C * plane), 0, (int)((uint32_t)C * plane * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(idx + n * (int64_t)C * plane), 0, (int)((uint32_t)C * plane), 0x00020000);
    uint32_t cell[CPT];          // cell index inside the plane; 0x20000000 = outside (reads 0 through the descriptors)
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int id = k * kBlock + threadIdx.x, trow = id / Wp, col = id - trow * Wp, r = t0 - 1 + trow;
        cell[k] = (id < cells && r >= 0 && r < Ho && col < Wo) ? (uint32_t)(r * Wo + col) : 0x20000000u;
    }
    f32x2 acc[CPT][18];
#pragma unroll
    for (int k = 0; k < CPT; ++k)
#pragma unroll
        for (int j = 0; j < 18; ++j) acc[k][j] = (f32x2){0.0f, 0.0f};

    auto request = [&](uint32_t c, float (&g)[CPT], uint32_t (&code)[CPT]) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            code[k] = __builtin_amdgcn_raw_buffer_load_b8(ir, cell[k], c * plane, 0);
            g[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, cell[k] * 4u, c * plane * 4u, 0));
        }
    };
    auto accumulate = [&](int c, const float (&g)[CPT], const uint32_t (&code)[CPT]) {
        const uint32_t cbase = (uint32_t)c * kChanBytes;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const float4 *w4 = reinterpret_cast<const float4 *>(tab + ((code[k] & 7u) * (uint32_t)kWinBytes + cbase));
            const f32x2 gg = {g[k], g[k]};
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float4 a = w4[j];
                acc[k][2 * j] = __builtin_elementwise_fma(gg, (f32x2){a.x, a.y}, acc[k][2 * j]);
                acc[k][2 * j + 1] = __builtin_elementwise_fma(gg, (f32x2){a.z, a.w}, acc[k][2 * j + 1]);
            }
        }
    };
    // two channels per trip, two register sets: the next channel's (code, gradient) pairs are in flight under this one's products
    float ga[CPT], gb[CPT];
    uint32_t ca[CPT], cb[CPT];
    request(0, ga, ca);
    for (int c = 0; c < C; c += 2) {
        const bool second = c + 1 < C;
        request((uint32_t)(second ? c + 1 : c), gb, cb);
        accumulate(c, ga, ca);
        if (second) {
            request((uint32_t)(c + 2 < C ? c + 2 : c), ga, ca);
            accumulate(c + 1, gb, cb);
        }
    }
    // the exchange: block (bh, bw) of a cell's window (rows 2 bh, 2 bh + 1; columns 2 bw, 2 bw + 1) belongs to patch
    // (row + bh - 1, col + bw - 1)
    float4 pacc[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) pacc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // three phases (block rows), three float4 per cell and phase; the buffer lies over the window tables, which nobody reads
    // any more (one tile per workgroup)
    __syncthreads();
#pragma unroll
    for (int bh = 0; bh < 3; ++bh) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int id = k * kBlock + threadIdx.x;
            if (id < cells) {
#pragma unroll
                for (int bw = 0; bw < 3; ++bw) {
                    const f32x2 top = acc[k][(2 * bh) * 3 + bw], bot = acc[k][(2 * bh + 1) * 3 + bw];
                    xch[bw * cells + id] = make_float4(top.x, top.y, bot.x, bot.y);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int id = k * kBlock + threadIdx.x, trow = id / Wp, col = id - trow * Wp;
            const int sr = trow + 1 - bh;          // source cell's tile row
            if (id < cells && sr >= 0 && sr < TR) {
#pragma unroll
                for (int bw = 0; bw < 3; ++bw) {
                    const int sc = col + 1 - bw;
                    if (sc >= 0 && sc < Wp) {
                        const float4 v = xch[bw * cells + sr * Wp + sc];
                        pacc[k].x += v.x;
                        pacc[k].y += v.y;
                        pacc[k].z += v.z;
                        pacc[k].w += v.w;
                    }
                }
            }
        }
        if (bh < 2) __syncthreads();
    }
    float *xn = gx + n * (int64_t)H * W;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int id = k * kBlock + threadIdx.x, trow = id / Wp, col = id - trow * Wp, hp = t0 - 1 + trow;
        if (id < cells && trow >= 1 && trow <= TR - 2 && hp < Hp) {
            const int h0 = 2 * hp, w0 = 2 * col;
            *reinterpret_cast<f32x2 *>(xn + (int64_t)h0 * W + w0) = (f32x2){pacc[k].x, pacc[k].y};
            if (h0 + 1 < H) *reinterpret_cast<f32x2 *>(xn + (int64_t)(h0 + 1) * W + w0) = (f32x2){pacc[k].z, pacc[k].w};
        }
    }
}

// ADVSTEP_CONV0_BWD=gather keeps the patch-centric kernel (A/B measurements); read at every call
inline bool conv0_cells_enabled() {
    const char *e = getenv("ADVSTEP_CONV0_BWD");
    return !(e && e[0] == 'g');
}

constexpr int64_t kMaxGridY = 65535;

}  // namespace

#define CONV0_REQUIRE(cond) \
    do {                    \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

int advstep_conv5_mfm_pool2_forward_f32(const float *x, const float *weight, const float *bias, float *y, uint8_t *idx,
                                        int64_t N, int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    CONV0_REQUIRE(N >= 0 && C >= 0 && H >= 0 && W >= 0);
    const int64_t Ho = H / 2, Wo = W / 2;
    if (N == 0 || C == 0 || Ho == 0 || Wo == 0) return ADVSTEP_OK;
    CONV0_REQUIRE(x && weight && y && idx && N <= kMaxGridY && C <= 4096 && H * W <= INT32_MAX);
    const dim3 grid((unsigned)ceil_div(Ho * Wo, kBlock), (unsigned)N);
    hipLaunchKernelGGL(conv5_mfm_pool2_forward_kernel, grid, dim3(kBlock), 0, as_stream(stream), x, weight, bias, y, idx,
                       (int)C, (int)H, (int)W);
    return status_after_launch();
}

int advstep_conv5_mfm_pool2_backward_f32(const float *gy, const uint8_t *idx, const float *weight, float *gx, int64_t N,
                                         int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    CONV0_REQUIRE(N >= 0 && C >= 0 && H >= 0 && W >= 0);
    if (N == 0 || H == 0 || W == 0) return ADVSTEP_OK;
    CONV0_REQUIRE(gx && N <= kMaxGridY && C <= 96 && H * W <= INT32_MAX);   // 2 C bordered 7x7 tables must fit 64 KB of LDS
    hipStream_t st = as_stream(stream);
    if (C == 0 || H / 2 == 0 || W / 2 == 0)
        return hipMemsetAsync(gx, 0, (size_t)N * H * W * sizeof(float), st) == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH;
    CONV0_REQUIRE(gy && idx && weight);
    const int64_t Wp = (W + 1) / 2, Hp = (H + 1) / 2;
    if (conv0_cells_enabled() && Wp <= kCellsMaxWidth && W % 2 == 0 && (reinterpret_cast<uintptr_t>(gx) & 7u) == 0 &&
        C * (H / 2) * (W / 2) < (1 << 27)) {
        // cell-centric kernel: full-width tiles of TR = 3 * 256 / Wp rows (two of them halo), trimmed to what the tile count needs.
        // (At LCNN's 202 x 40 cells and B = 128: 12 tiles of 17 + 2 rows per sample = 1 536 workgroups, exactly two rounds of the
        // three workgroups a compute unit holds; 2 or 4 cells per thread were measured slower: 80 / 101 us against 70.)
        constexpr int CPT = 3;
        const int TRmax = (int)((int64_t)CPT * kBlock / Wp);
        const int tiles = TRmax >= 3 ? (int)ceil_div(Hp, TRmax - 2) : 0;
        const int TR = tiles ? (int)ceil_div(Hp, tiles) + 2 : 0;
        const size_t xch = (size_t)3 * TR * Wp * sizeof(float4), tabs = (size_t)C * kChanBytes;
        const size_t lds = xch > tabs ? xch : tabs;
        // The opt-in to more than 64 KB of dynamic LDS is made once per device (function attributes are per device) and its
        // RESULT kept: 160 KB when the runtime granted it, else the 64 KB every kernel has.  Always the same value, so two
        // host threads launching different channel counts cannot undercut each other.  Shapes whose tables need more than
        // the limit fall through to the gather kernel below (< 38 KB), which handled them in rounds 1-4 (a refused
        // attribute used to end in ADVSTEP_ELAUNCH, and the attribute call ran at every launch).
        static std::atomic<int> granted_kb[64];                       // 0 = not asked yet
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<int> &slot = granted_kb[dev & 63];
        int kb = slot.load(std::memory_order_relaxed);
        if (kb == 0) {
            kb = hipFuncSetAttribute((const void *)conv5_mfm_pool2_backward_cells_kernel<CPT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 160 : 64;
            (void)hipGetLastError();                                  // a refusal is handled here, not reported by the launch below
            slot.store(kb, std::memory_order_relaxed);
        }
        const size_t cells_lds_limit = (size_t)kb * 1024;
        if (tiles && lds <= cells_lds_limit) {
            hipLaunchKernelGGL(conv5_mfm_pool2_backward_cells_kernel<CPT>, dim3((unsigned)tiles, (unsigned)N), dim3(kBlock), lds, st,
                               gy, idx, weight, gx, (int)C, (int)H, (int)W, TR);
            return status_after_launch();
        }
    }
    const int64_t patches = Hp * Wp;
    const dim3 grid((unsigned)ceil_div(patches, kBlock), (unsigned)N);
    const size_t lds = (size_t)(2 * (C * kTabWords + 4) + 8) * sizeof(float);
    hipLaunchKernelGGL(conv5_mfm_pool2_backward_kernel, grid, dim3(kBlock), lds, st, gy, idx, weight, gx, (int)C, (int)H,
                       (int)W);
    return status_after_launch();
}

}  // extern "C"
