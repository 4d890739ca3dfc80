// lcnn_wino.hip — LCNN's 3x3 convolution blocks on the fp32 matrix cores (C ABI: include/advstep_lcnn.h):
//     Conv2d(Cin, 2C, (3, 3), padding 1) -> MaxFeatureMap2D -> MaxPool2d(2, 2) [-> BatchNorm2d(eval, affine=False)]
//     (src/models/lcnn.py:128-131, 135-137, 149-154)  as ONE kernel, and the matching input-gradient convolution.
//
// These blocks are where an attack iteration spends its time (2 x 93 GFLOP per iteration at B = 128).  MIOpen runs them
// as a VALU Winograd kernel (miopenSp3AsmConv..f2x3, ~98 TFLOP/s direct-conv equivalent) whose 2C-channel output is
// written, re-read by the max-feature-map + pool kernel, and written again.  Here: Winograd F(2x2, 3x3) with the 16
// per-position GEMMs on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate), max-feature-map + pool + BatchNorm in
// the epilogue, so the conv output never exists.
//
//   * wave tile: 16 Winograd tiles (2x2 outputs each) x 32 output channels x all 16 positions = 128 accumulator
//     registers.  The matrix instruction's B operand — V[xi][cin][tile] — is computed by the lane that feeds it: lane
//     (g, n) loads the 4x4 input patch of (cin = 4 s + g, tile n) and applies B^T d B in registers (32 adds), which
//     yields exactly its B values for all 16 positions of k-step s.  No LDS round trip for the input.
//   * zero padding comes from the buffer descriptor: out-of-image taps get an out-of-range offset and read as 0.
//   * the transformed weights U = G g G^T (prepared once per weight version, advstep_conv3x3_prepare_f32) stream
//     through LDS in 16-channel chunks laid out so one ds_read_b64 fetches a lane's A values for both accumulator
//     tiles; K <= 64 stays resident, larger K (the input-gradient convolutions, K = 2C) is double-buffered.
//   * forward: the 32 channels of a workgroup are 16 max-feature-map PAIRS (c, c + C): both halves of a pair land in
//     the same lane and register slot, and a Winograd tile is exactly one 2x2 pooling window, so
//     bias -> pair max -> pool -> BatchNorm is register-only; selection goes out as one byte per pooled output (the
//     format of advstep_mfm_pool2_forward_f32, consumed by advstep_mfm_pool2_backward_f32).
//   * input gradient: the same kernel with U built from the rotated, transposed weights and a plain 2x2 store.
// Winograd in fp32 rounds differently from a direct convolution (as MIOpen's own Winograd does): tests compare against
// ATen / MIOpen with a tolerance, selection flips are counted.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "advstep_detector.h"
#include "advstep_lcnn.h"

namespace {

constexpr int kWaves = 8, kThreads = kWaves * 64;
constexpr int kChunkCin = 16;                         // input channels per LDS chunk
constexpr int kChunkFloats = 16 * kChunkCin * 32;     // [xi][cin][16 j][2 m] = 32 KB
constexpr int kMaxResident = 4;                       // chunks kept in LDS when K <= 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

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

// ---- weight transform: U = G g G^T into the chunked layout ------------------------------------------------------------
// mode 0 (forward, max-feature-map pairs): output row (slice, j, m) = conv channel m * C + slice * 16 + j, g = weight.
// mode 1 (input gradient): the convolution that maps d(conv out) (2C channels) to d(conv in) (Cin channels) has kernel
//         g'[ci][co][a][b] = weight[co][ci][2 - a][2 - b]; output row (slice, j, m) = input channel slice*32 + m*16 + j.
// mode 2: mode 1 with the reduction channels in the compact source's order (halves interleaved per k-step, see below).
// U: [slice][chunk][xi][16 cin][16 j][2 m], zero where a row / channel does not exist.
__global__ void wino_prepare_kernel(const float *__restrict__ weight, const float *__restrict__ kscale,
                                    float *__restrict__ U, int Cin, int Cout, int mode, int slices, int chunks) {
    const int K = mode == 0 ? Cin : Cout;        // reduction channels
    const int total = slices * 32 * K;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = i % K, row = i / K, slice = row / 32, jm = row % 32, j = jm % 16, m = jm / 16;
    float g[3][3];
    bool live;
    if (mode == 0) {
        const int C = Cout / 2, c = slice * 16 + j;
        live = c < C;
        const int co = m * C + c;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = live ? weight[((int64_t)co * Cin + k) * 9 + a * 3 + b] : 0.0f;
    } else {
        // mode 2 (the compact source of advstep_conv3x3_mfm_pool2_backward_f32): the reduction runs over the two halves of a
        // max-feature-map channel in ADJACENT k-steps — k-step 2 j holds channels 4 j .. 4 j + 3 of the first half, k-step
        // 2 j + 1 the same channels of the second half — so one load of the pooled gradients + selection bytes serves both
        const int kk = mode == 2 ? ((k >> 2) & 1) * (Cout / 2) + (k >> 3) * 4 + (k & 3) : k;
        const int ci = slice * 32 + m * 16 + j;
        live = ci < Cin;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = live ? weight[((int64_t)kk * Cin + ci) * 9 + (2 - a) * 3 + (2 - b)] : 0.0f;
        // a per-channel factor on d(conv out) — the folded BatchNorm's invstd of max-feature-map channel kk % C — is a
        // factor on row k of the operand
        if (kscale) {
            const float f = kscale[kk % (Cout / 2)];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] *= f;
        }
    }
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[4][3], u[4][4];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        u[a][0] = t[a][0];
        u[a][1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        u[a][2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        u[a][3] = t[a][2];
    }
    const int chunk = k / kChunkCin, kc = k % kChunkCin;
    float *dst = U + ((int64_t)(slice * chunks + chunk)) * kChunkFloats + (kc * 16 + j) * 2 + m;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) dst[xi * (kChunkCin * 32)] = u[xi >> 2][xi & 3];
}

// Plain convolutions (advstep_resconv_*): R output rows, reduction over K1 channels with 3x3 taps followed by K2 channels
// with a centre tap only (a 1x1 convolution of a second tensor).  transpose = 0: w3 (R, K1, 3, 3), w1 (R, K2).
// transpose = 1 (input gradient of a convolution with weight w3 (K1, R, 3, 3) [and w1 (K2, R)]): taps rotated, channels swapped.
// rscale (R) multiplies a row, kscale (K1) the 3x3 part's reduction channel.  Row (slice, j, m) = slice * 32 + m * 16 + j.
__global__ void wino_prepare_plain_kernel(const float *__restrict__ w3, const float *__restrict__ w1,
                                          const float *__restrict__ rscale, const float *__restrict__ kscale,
                                          float *__restrict__ U, int R, int K1, int K2, int transpose, int slices, int chunks) {
    const int Kp = chunks * kChunkCin;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slices * 32 * Kp) return;
    const int k = i % Kp, rowi = i / Kp, slice = rowi / 32, jm = rowi % 32, j = jm % 16, m = jm / 16;
    const int row = slice * 32 + m * 16 + j;
    float g[3][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    if (row < R && k < K1) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                g[a][b] = transpose ? w3[((int64_t)k * R + row) * 9 + (2 - a) * 3 + (2 - b)] : w3[((int64_t)row * K1 + k) * 9 + a * 3 + b];
        if (kscale) {
            const float f = kscale[k];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] *= f;
        }
    } else if (row < R && k < K1 + K2) {
        g[1][1] = transpose ? w1[(int64_t)(k - K1) * R + row] : w1[(int64_t)row * K2 + (k - K1)];
    }
    if (rscale && row < R) {
        const float f = rscale[row];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] *= f;
    }
    float t[4][3], u[4][4];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        u[a][0] = t[a][0];
        u[a][1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        u[a][2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        u[a][3] = t[a][2];
    }
    const int chunk = k / kChunkCin, kc = k % kChunkCin;
    float *dst = U + ((int64_t)(slice * chunks + chunk)) * kChunkFloats + (kc * 16 + j) * 2 + m;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) dst[xi * (kChunkCin * 32)] = u[xi >> 2][xi & 3];
}

// ---- the convolution ---------------------------------------------------------------------------------------------------
// EPI 0: plain store of min(32, Cout - slice * 32) channels per slice (the input-gradient convolution).
// EPI 1: bias + max-feature-map + 2x2 pool [+ BatchNorm]; Cout = number of max-feature-map channels C.
// EPI 2: bias + max-feature-map [+ BatchNorm] without the pool; one selection byte per 2x2 tile.
// STREAM: K > 64, U chunks streamed through LDS (two buffers and a barrier per chunk; SRC 1: four buffers, a barrier per two
//         chunks, one stream across tile groups); otherwise all chunks stay resident.
// grid = slices * ranges workgroups; workgroup b: slice b % slices, tile range b / slices.
// SRC 0: the input is a dense tensor x (N, K, H, W).
// SRC 1: the input is d(conv out) of a max-feature-map + 2x2 pool block, given in its compact form — the pooled gradient
//        gy (N, C, H/2, W/2) and the selection bytes (advstep_mfm_pool2_forward_f32's encoding), K = 2C: channel k of
//        half k / C at conv position (h, w) carries gy[k % C][h/2][w/2] if that position of that half won, else 0.  The
//        4x4 patch of a lane is expanded from the 3x3 pooled cells around its tile; the dense gradient never exists.
// SRC 2: the same for a plain MaxPool2d(2) (no halves): gy (N, K, H/2, W/2) and ATen-order selection bytes (2 * dh + dw):
//        channel k at (h, w) carries gy[k][h/2][w/2] if that position won its window, else 0 (odd trailing row / column: 0).
// EPI 3: + shift[ch], LeakyReLU(slope), plain store            (the residual blocks of SpecRNet, advstep_detector.h)
// EPI 4: + bias[ch], MaxPool2d(2) with ATen's selection byte (detector_elem.hip::pool4): the conv output never exists.
// EPI 5: * (h > 0 ? 1 : slope) with h (N, Cout, H, W) passed in `bn_mean` — LeakyReLU's backward from its OUTPUT (same sign as
//        its input for slope > 0) — plain store.
// EPI 6: the same from the activation's SIGN BYTES passed in `idx` (read only): one byte per (n, channel, 2x2 tile), bit 2 i + j =
//        h > 0 at tile position (i, j), (N, Cout, TH, TW) — what EPI 3 writes next to h when it is given an `idx` to fill.  At
//        SpecRNet's first block that is 21 MB in the epilogue instead of 331 MB of h: 476 -> 3xx us (round 3).
// EPI 7: EPI 4 plus a 1x1 convolution over the ga.few (1 or 2) channels of ga.x2 added to the convolution output on the vector
//        ALUs before the pool, weights (Cout, few) in `bn_mean`: SpecRNet's block0, whose downsample convolution has two input
//        channels — as part of the reduction (GEN's x2) those two channels cost a whole k-step of six, 32 matrix instructions per
//        wave and tile group for 8 useful rows of K; here they are 64 fused multiply-adds per lane.
// NT: accumulator tiles a wave computes — 2, or 1 for a convolution's LAST slice when only its first 16 rows exist
//     (Cout % 32 in 1..16, LCNN's 128 -> 48 input gradient): that slice is launched on its own with half the matrix
//     instructions instead of multiplying 16 zero rows.  slice0: first slice of this launch.
// GEN (SRC 0 only): the reduction runs over the channels of TWO dense tensors, x (K1 channels, K1 % 4 == 0 when x2 is
//     given) then x2 (Kreal - K1 channels) — a 1x1 convolution of x2 added to the 3x3 convolution of x is the same
//     reduction with centre-tap-only weights — and Kreal need not fill the last k-step: K is Kreal rounded up to 4 (>= 8),
//     channels >= Kreal get an out-of-range offset (read 0; their U rows are 0 as well).
struct GenArgs {
    const float *x2;
    int K1, Kreal;
    float slope;
    int xcd;          // 1: workgroups b, b + 8, b + 16, ... (one XCD, one L2) take the slices of the same tile ranges
    int few;          // EPI 7: channels of x2 (1 or 2) whose 1x1 convolution is added in the epilogue
    int halves;       // NT == 1, round 5: the launch's "slices" are the two 16-row HALVES of slice `slice0` (rows 16 h .. 16 h + 15:
                      // component h of the A pairs) - a one-slice layer on a small plane (LCNN's 64 -> 32 input gradient at 50 x 10:
                      // 1 000 tile groups, 125 workgroups) then fills the chip with 250 workgroups of half the matrix instructions
};

// WODD (SRC 0): the plane width is odd, so the last tile of a row has no second column and its pair load's second element must
//     be masked; even widths (all of LCNN's) skip those 4 selects per k-step.
template <int EPI, bool STREAM, int SRC, int NT = 2, bool GEN = false, bool WODD = false>
__global__ __launch_bounds__(kThreads) void wino3x3_kernel(const float *__restrict__ x, const uint8_t *__restrict__ xsel,
                                                           const float *__restrict__ U,
                                                           const float *__restrict__ bias,
                                                           const float *__restrict__ bn_mean,
                                                           const float *__restrict__ bn_invstd, float *__restrict__ y,
                                                           uint8_t *__restrict__ idx, int N, int K, int H, int W, int Cout,
                                                           int slices, int ranges, int slice0, GenArgs ga) {
    static_assert(NT == 2 || EPI == 0 || EPI == 3 || EPI == 5 || EPI == 6, "one accumulator tile only for the plain-store epilogues");
    static_assert(!GEN || SRC != 1, "the general reduction reads dense tensors or a plain pooled gradient");
    extern __shared__ __attribute__((aligned(16))) float u_s[];
    int slice_i = blockIdx.x % slices, range = blockIdx.x / slices;
    if (ga.xcd) {
        const int x = blockIdx.x & 7, w = blockIdx.x >> 3;
        slice_i = w % slices;
        range = x * (ranges >> 3) + w / slices;
    }
    const int half = (NT == 1 && ga.halves) ? slice_i : 0;
    const int slice = (NT == 1 && ga.halves) ? slice0 : slice0 + slice_i;
    const int chunks = (K + kChunkCin - 1) / kChunkCin, steps = K / 4;
    const float *Usl = U + (int64_t)slice * chunks * kChunkFloats;
    // STREAM with a compact source: a chunk goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, nothing
    // to wait for at the point of issue; round 3).  Through registers the four loads of a chunk are followed at once by their four
    // ds_write, i.e. by s_waitcnt vmcnt(0), right after every chunk barrier.  Measured: compact-source input gradients 286 -> 281
    // and 170 -> 167 us; the dense-source ones get SLOWER (276 -> 287 us: with a DMA in flight hipcc waits vmcnt(0) at the next
    // use of an ordinary load, which drains their patch prefetch), so they keep the register path.  The DMA's completion is waited
    // for (vmcnt(0)) before the barrier that publishes the chunk.  The LDS destination of a wave-instruction is a wave-uniform
    // base + lane x 16 B, which is exactly this copy's layout.
#ifndef WINO_GLDS
#define WINO_GLDS (SRC == 1)
#endif
    auto copy_chunk = [&](int chunk, int buf) {
        const float4 *src = reinterpret_cast<const float4 *>(Usl + (int64_t)chunk * kChunkFloats);
        float4 *dst = reinterpret_cast<float4 *>(u_s + buf * kChunkFloats);
        if (STREAM && WINO_GLDS) {
            const int w64 = (threadIdx.x >> 6) << 6;
#pragma unroll
            for (int i = 0; i < kChunkFloats / 4 / kThreads; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + threadIdx.x + i * kThreads),
                                                 (__attribute__((address_space(3))) void *)(dst + w64 + i * kThreads), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < kChunkFloats / 4 / kThreads; ++i) dst[threadIdx.x + i * kThreads] = src[threadIdx.x + i * kThreads];
        }
    };
    // every LDS-DMA transfer of this wave has landed (vmcnt(0); expcnt / lgkmcnt untouched) — before the barrier that publishes it
    auto dma_landed = [&]() {
        if (STREAM && WINO_GLDS) __builtin_amdgcn_s_waitcnt(0x0F70);
    };
    // Per-slice epilogue constants, staged once per workgroup (round 3; they were 16 global loads + their address arithmetic per
    // tile group): cst[0..15] / cst[16..31] = what is added to the two accumulator tiles' rows — the convolution's bias (EPI 1 / 2:
    // of the two max-feature-map halves, minus the folded BatchNorm's mean: max-feature-map and max-pool commute with a
    // per-channel shift), EPI 3's shift, EPI 4's bias; cst[32..47] = the BatchNorm's 1 / std (EPI 1 / 2).  The additive part
    // enters through the matrix instruction's C operand: position (1, 1) of M reaches all four outputs of A^T M A with weight
    // +1, so its accumulator starts at the constant instead of 0 and the epilogue has no bias adds.
    constexpr bool kHasConst = EPI == 1 || EPI == 2 || EPI == 3 || EPI == 4 || EPI == 7;
    __shared__ __attribute__((aligned(16))) float cst[48];
    __shared__ float few_w[EPI == 7 ? 64 : 1];               // EPI 7: the 1x1 weights of this slice's 32 rows, [row][channel]
    if (EPI == 7 && threadIdx.x >= 64 && threadIdx.x < 128) {
        const int i = threadIdx.x - 64, row = i >> 1, c = i & 1, ch = slice * 32 + row;
        few_w[i] = (ch < Cout && c < ga.few) ? bn_mean[ch * ga.few + c] : 0.0f;
    }
    if (kHasConst && threadIdx.x < 48) {
        const int q = threadIdx.x >> 4, j = threadIdx.x & 15;
        float v = q == 2 ? 1.0f : 0.0f;
        if (EPI == 1 || EPI == 2) {
            const int ch = slice * 16 + j;
            if (ch < Cout) {
                if (q == 2) v = bn_mean ? bn_invstd[ch] : 1.0f;
                else v = (bias ? bias[ch + q * Cout] : 0.0f) - (bn_mean ? bn_mean[ch] : 0.0f);
            }
        } else if (q < 2) {
            const int ch = slice * 32 + q * 16 + j;
            if (ch < Cout && bias) v = bias[ch];
        }
        cst[threadIdx.x] = v;
    }
    if (!STREAM) {
        for (int c = 0; c < chunks; ++c) copy_chunk(c, c);
    }
    if (!STREAM || kHasConst) __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, nl = lane & 15;
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    const int tiles = N * TH * TW, groups = (tiles + 15) >> 4;
    const int iters = (groups + ranges * kWaves - 1) / (ranges * kWaves);   // the same for every workgroup
    const uint32_t plane = (uint32_t)(H * W);
    // raw buffer over x: an out-of-range offset reads as 0 — the convolution's zero padding, for free
    // SRC 1: pooled grid, max-feature-map channels (two conv channels per pooled channel); SRC 2: pooled grid, Kreal channels
    const int Hs = H >> 1, Ws = W >> 1, Cs = SRC == 1 ? K >> 1 : (GEN ? ga.Kreal : K);
    const uint32_t cplane = (uint32_t)(Hs * Ws);
    const int KA = GEN ? ga.K1 : K, KB = GEN ? ga.Kreal - ga.K1 : 0;      // channels of x and of x2
    const size_t src_elems = SRC == 0 ? (size_t)N * KA * plane : (size_t)N * Cs * cplane;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)(src_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(GEN && KB > 0 ? ga.x2 : x), 0, GEN && KB > 0 ? (int)((size_t)N * KB * plane * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t sr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(SRC != 0 ? xsel : reinterpret_cast<const uint8_t *>(x)), 0,
                                          (int)src_elems, 0x00020000);

    if (STREAM && SRC == 1) {       // the U stream's first two chunks (see the k loop of SRC 1)
        copy_chunk(0, 0);
        copy_chunk(1, 1);
    }
    for (int it = 0; it < iters; ++it) {
        const int grp = (it * ranges + range) * kWaves + wave;
        const int t = grp * 16 + nl;
        const bool valid = grp < groups && t < tiles;
        const int tt = valid ? t : 0;
        const int n = tt / (TH * TW), rem = tt - n * (TH * TW), th = rem / TW, tw = rem - th * TW;
        // patch addressing: one lane base + compile-time (p, q) strides; the 16 validity flags live in scalar registers
        // as lane masks, and an invalid tap gets an out-of-range offset (reads 0) when the load is issued
        const uint32_t lane_base = (((uint32_t)(n * KA + g)) * plane + (uint32_t)((2 * th - 1) * W + (2 * tw - 1))) * 4u;
        const uint32_t lane_base2 = (((uint32_t)(n * KB + g)) * plane + (uint32_t)((2 * th - 1) * W + (2 * tw - 1))) * 4u;
        // SRC 0: a patch row is [column 2 tw - 1 | columns 2 tw, 2 tw + 1 | column 2 tw + 2].  The middle pair is ONE 8-byte load
        // (the 16 lanes of a lane row hold consecutive tiles: 128 contiguous bytes); the outer columns are the neighbouring
        // tiles' pairs and come from the neighbouring LANES (DPP row shift) — only the first / last lane of a lane row loads its
        // outer column itself (one dword load per patch row with 2 of 16 lanes active).  8 load instructions per k-step instead of
        // 16 with lanes 8 bytes apart: the patch loads, not the matrix pipe, were what a quarter of these kernels' time went to
        // (halving them — wrong results, timing only — made the L6 / L13 kernels 9-25 % faster).
        bool row_ok[4], ok_l[4], ok_r[4], ok_2[4], ok_e[4];      // lane masks (scalar registers)
        uint32_t pair_off[4], edge_off[4];                       // byte offsets from the lane base (GEN) / the tensor base
        uint32_t cell[3][3];     // SRC 1: element offsets of the 3x3 pooled cells around the tile (0x20000000 = outside)
        if (SRC == 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int hh = 2 * th - 1 + p;
                row_ok[p] = valid && hh >= 0 && hh < H;
                ok_l[p] = row_ok[p] && tw > 0;                   // column 2 tw - 1 exists
                ok_2[p] = row_ok[p] && 2 * tw + 1 < W;           // second element of the pair exists (odd W: not in the last tile)
                ok_r[p] = row_ok[p] && 2 * tw + 2 < W;           // column 2 tw + 2 exists
                ok_e[p] = (nl == 0 && ok_l[p]) || (nl == 15 && ok_r[p]);
                const uint32_t rel_pair = (uint32_t)((p * W + 1) * 4), rel_edge = (uint32_t)((p * W + (nl == 0 ? 0 : 3)) * 4);
                pair_off[p] = GEN ? rel_pair : (row_ok[p] ? lane_base + rel_pair : 0x80000000u);
                edge_off[p] = GEN ? rel_edge : (ok_e[p] ? lane_base + rel_edge : 0x80000000u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ci = th - 1 + i, cj = tw - 1 + j;
                    const bool in = valid && ci >= 0 && ci < Hs && cj >= 0 && cj < Ws;
                    cell[i][j] = in ? ((uint32_t)(n * Cs + g)) * cplane + (uint32_t)(ci * Ws + cj) : 0x20000000u;
                }
        }
        f32x4 acc[16][NT];

        // a patch in flight: SRC 0 the 16 taps; SRC 1 the 9 pooled gradients + their 9 selection bytes
        struct Patch {
            float v[SRC == 0 ? 12 : 9];          // SRC 0: per patch row the pair (2 p, 2 p + 1), then the 4 edge values (8 + p)
            uint32_t code[SRC == 0 ? 1 : 9];
        };
        auto load_row = [&](Patch &dst, int p, __amdgpu_buffer_rsrc_t r, uint32_t po, uint32_t eo, uint32_t soff) {
            const f32x2 pr = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, po, soff, 0));
            dst.v[2 * p] = pr.x;
            dst.v[2 * p + 1] = pr.y;
            dst.v[8 + p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, eo, soff, 0));
        };
        auto load_patch = [&](Patch &dst, int s) {
            if (SRC == 0 && GEN) {
                const bool first = 4 * s < ga.K1;                          // wave-uniform: this k-step reads x (else x2)
                const uint32_t soff = (uint32_t)(first ? 4 * s : 4 * s - ga.K1) * plane * 4u;
                const bool lane_ok = 4 * s + g < ga.Kreal;                // padded reduction channels read 0
                const uint32_t base = first ? lane_base : lane_base2;
                const __amdgpu_buffer_rsrc_t r = first ? xr : xr2;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    load_row(dst, p, r, (row_ok[p] && lane_ok) ? base + pair_off[p] : 0x80000000u,
                             (ok_e[p] && lane_ok) ? base + edge_off[p] : 0x80000000u, soff);
            } else if (SRC == 0) {
                const uint32_t soff = (uint32_t)(4 * s) * plane * 4u;
#pragma unroll
                for (int p = 0; p < 4; ++p) load_row(dst, p, xr, pair_off[p], edge_off[p], soff);
            } else {
                const int k0 = 4 * s, c0 = k0;      // wave-uniform: channel of lane group 0 (SRC 1: `s` counts k-step PAIRS)
                const uint32_t soff = (uint32_t)c0 * cplane;
                const bool lane_ok = SRC == 1 || !GEN || k0 + g < ga.Kreal;        // padded reduction channels read 0
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const uint32_t e = lane_ok ? cell[i / 3][i % 3] : 0x20000000u;
                    dst.v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, e << 2, soff << 2, 0));
                    dst.code[i] = __builtin_amdgcn_raw_buffer_load_b8(sr, e, soff, 0);
                }
            }
        };
        // the 4x4 taps of a patch: SRC 1 routes each pooled gradient to the one conv position (of the one half) that won
        auto taps = [&](const Patch &src, int s, float (&d)[4][4]) {
            if (SRC == 0) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float a = src.v[2 * p], b = src.v[2 * p + 1], e = src.v[8 + p];
                    // row_shr:1 / row_shl:1 inside the 16-lane row; a lane without a source (the row's first / last) keeps `e`
                    const float left = __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, b), 0x111, 0xf, 0xf, false));
                    const float right = __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, a), 0x101, 0xf, 0xf, false));
                    d[p][0] = ok_l[p] ? left : 0.0f;
                    d[p][1] = a;
                    d[p][2] = (WODD && !ok_2[p]) ? 0.0f : b;      // even width: an invalid row's pair was loaded as 0 already
                    d[p][3] = ok_r[p] ? right : 0.0f;
                }
            } else {
                const uint32_t half_bit = SRC == 1 && (s & 1) ? 4u : 0u;              // wave-uniform: odd k-steps are the second half
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = ((p + 1) >> 1) * 3 + ((q + 1) >> 1);
                        const uint32_t want = half_bit | (uint32_t)((((p + 1) & 1) << 1) | ((q + 1) & 1));
                        d[p][q] = src.code[i] == want ? src.v[i] : 0.0f;
                    }
            }
        };
        // one k-step.  The A operands come from LDS in 8 groups (2 Winograd positions x 2 accumulator tiles = one
        // ds_read2_b64 each); group i + 2 is requested before the 4 matrix instructions of group i issue, and the input
        // transform V = B^T d B (32 adds) runs behind the first two requests: the LDS latency is never exposed.  The
        // scheduling barriers pin that order (left alone, the compiler sinks every read next to its use and waits).
        auto step = [&](const Patch &patch, int s_idx, const float *us, auto first) {
            f32x2 a[8][2];
            auto request = [&](int grp) {
                if constexpr (NT == 1) {      // one accumulator tile: only this half's component of the pair (4-byte reads)
                    a[grp][0].x = us[(2 * grp) * (kChunkCin * 32) + half];
                    a[grp][1].x = us[(2 * grp + 1) * (kChunkCin * 32) + half];
                } else {
                    a[grp][0] = *reinterpret_cast<const f32x2 *>(us + (2 * grp) * (kChunkCin * 32));
                    a[grp][1] = *reinterpret_cast<const f32x2 *>(us + (2 * grp + 1) * (kChunkCin * 32));
                }
            };
            request(0);
            request(1);
            __builtin_amdgcn_sched_barrier(0);
            float d[4][4], tr[4][4], v[4][4];
            taps(patch, s_idx, d);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tr[0][q] = d[0][q] - d[2][q];
                tr[1][q] = d[1][q] + d[2][q];
                tr[2][q] = d[2][q] - d[1][q];
                tr[3][q] = d[1][q] - d[3][q];
            }
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                v[a4][0] = tr[a4][0] - tr[a4][2];
                v[a4][1] = tr[a4][1] + tr[a4][2];
                v[a4][2] = tr[a4][2] - tr[a4][1];
                v[a4][3] = tr[a4][1] - tr[a4][3];
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            f32x4 c0 = zero, c1 = zero;                         // the first k-step's C operand of position (1, 1): see cst[]
            if (decltype(first)::value && kHasConst) {
                c0 = *reinterpret_cast<const f32x4 *>(cst + 4 * g);
                c1 = *reinterpret_cast<const f32x4 *>(cst + 16 + 4 * g);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                if (grp + 2 < 8) request(grp + 2);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int xi = 2 * grp + e;
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[grp][e].x, v[xi >> 2][xi & 3],
                                                                      decltype(first)::value ? (xi == 5 ? c0 : zero) : acc[xi][0], 0, 0, 0);
                    if (NT == 2)
                        acc[xi][NT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[grp][e].y, v[xi >> 2][xi & 3],
                                                                           decltype(first)::value ? (xi == 5 ? c1 : zero) : acc[xi][NT - 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        // lane's A address inside a chunk buffer for k-step s: cin_in_chunk = (s & 3) * 4 + g
        auto a_ptr = [&](int s, int buf) { return u_s + buf * kChunkFloats + (((s & 3) * 4 + g) * 16 + nl) * 2; };

        Patch da, db;
        if constexpr (SRC == 1) {
            // Compact max-feature-map source (round 3): the two halves (c, c + C) of a pooled channel read the SAME pooled gradient
            // and selection byte, so the reduction is ordered half 0 / half 1 of channels 4 j .. 4 j + 3 in k-steps 2 j / 2 j + 1
            // (U prepared with mode 2) and ONE patch serves two k-steps: half the vector-memory loads of the k loop.  `da` is
            // the pair in use, `db` the next pair, requested two k-steps before it is copied into `da` (18 moves per pair):
            // these kernels' loads come from the Infinity Cache / HBM (62 MB of pooled gradients + bytes per layer) and the one
            // k-step of distance they had did not cover them.  The loop body stays two k-steps long (a four-step body that
            // alternates two buffers without the moves spills 39 registers: 281 -> 310 us).
            // Round 4: with a streamed U (K > 64) the chunks run through FOUR LDS buffers as ONE stream across tile groups —
            // chunk q of the stream (q = it * chunks + c) lives in buffer q & 3 — with one workgroup barrier per TWO chunks: at
            // the barrier before an even q, chunks q and q + 1 (requested one barrier earlier) have landed for every wave, chunks
            // q - 2 and q - 1 have no reader left, and q + 2, q + 3 are requested into their buffers — the next tile group's
            // first two chunks when this one is ending.  Round 3 had two buffers, a barrier per chunk, and started every tile
            // group with an exposed request-wait-barrier for its chunk 0.  Timing-only builds without the per-chunk barrier ran
            // 6 - 9 % faster (8 waves, two per SIMD and deliberately out of phase, meet 6 - 8 times per tile group); half of the
            // meetings and the exposed start are what this removes.
            load_patch(da, 0);
            load_patch(db, 1);
            const int q0 = it * chunks;                     // even: K % 32 == 0 for these sources
            auto stream_point = [&](int c) {                // before chunk c (even) of this tile group
                dma_landed();           // vmcnt(0): BEFORE the next pair's patch is requested, or it would wait for that too
                __syncthreads();
#pragma unroll
                for (int d = 2; d < 4; ++d) {
                    const int cn = c + d;
                    if (cn < chunks) copy_chunk(cn, (q0 + cn) & 3);
                    else if (it + 1 < iters) copy_chunk(cn - chunks, (q0 + cn) & 3);
                }
            };
            if (STREAM) stream_point(0);
            step(da, 0, a_ptr(0, STREAM ? (q0 & 3) : 0), std::true_type{});
            step(da, 1, a_ptr(1, STREAM ? (q0 & 3) : 0), std::false_type{});
#pragma unroll 1
            for (int s = 2; s < steps; s += 2) {
                da = db;
                if (STREAM && (s & 7) == 0) stream_point(s >> 2);
                if (s + 2 < steps) load_patch(db, (s + 2) >> 1);
                const int buf = STREAM ? ((q0 + (s >> 2)) & 3) : (s >> 2);
                step(da, s, a_ptr(s, buf), std::false_type{});
                step(da, s + 1, a_ptr(s + 1, buf), std::false_type{});
            }
        } else {
        if (STREAM) {
            __syncthreads();            // previous iteration's readers are done with both buffers
            copy_chunk(0, 0);
            dma_landed();
            __syncthreads();
        }
        load_patch(da, 0);
        load_patch(db, 1);
        step(da, 0, a_ptr(0, 0), std::true_type{});
        load_patch(da, 2);
        step(db, 1, a_ptr(1, 0), std::false_type{});
        if (STREAM) copy_chunk(1, 1);   // chunks >= 2 always here; lands while chunk 0's last steps run
#pragma unroll 1
        for (int s = 2; s + 1 < steps; s += 2) {
            if (STREAM && (s & 3) == 0) {
                dma_landed();
                __syncthreads();        // chunk s/4 is complete in its buffer; chunk s/4 - 1 is free
                if (s / 4 + 1 < chunks) copy_chunk(s / 4 + 1, (s / 4 + 1) & 1);
            }
            const int buf = STREAM ? ((s >> 2) & 1) : (s >> 2);
            load_patch(db, s + 1);      // in flight while this step's matrix instructions run
            step(da, s, a_ptr(s, buf), std::false_type{});
            if (s + 2 < steps) load_patch(da, s + 2);
            step(db, s + 1, a_ptr(s + 1, buf), std::false_type{});
        }
        if (GEN && (steps & 1)) {
            // an odd number of k-steps (K a multiple of 4, not of 8 — SpecRNet's 20-channel layers: 5 steps instead of 6): the last
            // one on its own; its patch was requested by the last pair (or by the prologue when steps == 3)
            const int s = steps - 1;
            if (STREAM && (s & 3) == 0) {                       // its chunk was copied one pair earlier
                dma_landed();
                __syncthreads();
            }
            step(da, s, a_ptr(s, STREAM ? ((s >> 2) & 1) : (s >> 2)), std::false_type{});
        }
        }   // SRC != 1

        // epilogue: Y = A^T M A per (channel, tile)
        const int Ho = H >> 1, Wo = W >> 1;
        // EPI 5 reads the activation's output h at the tile's 2x2 positions of every row it stores — a 331 MB tensor at SpecRNet's
        // first block, i.e. HBM latency.  All of a lane's reads (4 rows x NT tiles x 2 tile rows) are requested HERE, before the
        // output transform's ~200 adds, instead of one row at a time right where they are used (round 3).
        float hv5[EPI == 5 ? 4 : 1][EPI == 5 ? NT : 1][2][2];
        if constexpr (EPI == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const int ch = slice * 32 + (m + half) * 16 + 4 * g + r;
                    const bool live = valid && ch < Cout;
                    const float *hp = bn_mean + (((size_t)n * Cout + (live ? ch : 0)) * H + 2 * th) * W + 2 * tw;
                    const bool h1 = 2 * th + 1 < H, w1 = 2 * tw + 1 < W;
                    hv5[r][m][0][0] = hv5[r][m][0][1] = hv5[r][m][1][0] = hv5[r][m][1][1] = 1.0f;
                    if (live) {
                        if ((W & 1) == 0) {
                            const f32x2 a = *reinterpret_cast<const f32x2 *>(hp);
                            hv5[r][m][0][0] = a.x, hv5[r][m][0][1] = a.y;
                            if (h1) {
                                const f32x2 b = *reinterpret_cast<const f32x2 *>(hp + W);
                                hv5[r][m][1][0] = b.x, hv5[r][m][1][1] = b.y;
                            }
                        } else {
                            hv5[r][m][0][0] = hp[0];
                            if (w1) hv5[r][m][0][1] = hp[1];
                            if (h1) {
                                hv5[r][m][1][0] = hp[W];
                                if (w1) hv5[r][m][1][1] = hp[W + 1];
                            }
                        }
                    }
                }
        }
        uint32_t hb6[EPI == 6 ? 4 : 1][EPI == 6 ? NT : 1];      // EPI 6: the same moment, one byte per (row, tile)
        if constexpr (EPI == 6) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const int ch = slice * 32 + (m + half) * 16 + 4 * g + r;
                    const bool live = valid && ch < Cout;
                    hb6[r][m] = live ? idx[(((size_t)n * Cout + ch) * TH + th) * TW + tw] : 0u;
                }
        }
        float xf[EPI == 7 ? 2 : 1][2][2];       // EPI 7: x2 at the tile's 2x2 positions, requested before the output transform
        if constexpr (EPI == 7) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                xf[c][0][0] = xf[c][0][1] = xf[c][1][0] = xf[c][1][1] = 0.0f;
                if (valid && c < ga.few && th < (H >> 1) && tw < (W >> 1)) {     // pooled outputs only: both rows / columns exist
                    const float *xp = ga.x2 + (((size_t)n * ga.few + c) * H + 2 * th) * W + 2 * tw;
                    if ((W & 1) == 0) {
                        const f32x2 a = *reinterpret_cast<const f32x2 *>(xp), b2 = *reinterpret_cast<const f32x2 *>(xp + W);
                        xf[c][0][0] = a.x, xf[c][0][1] = a.y, xf[c][1][0] = b2.x, xf[c][1][1] = b2.y;
                    } else {
                        xf[c][0][0] = xp[0], xf[c][0][1] = xp[1], xf[c][1][0] = xp[W], xf[c][1][1] = xp[W + 1];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float yy[2][2][2];
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                float s0[4], s1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    s0[b] = acc[0 + b][m][r] + acc[4 + b][m][r] + acc[8 + b][m][r];
                    s1[b] = acc[4 + b][m][r] - acc[8 + b][m][r] - acc[12 + b][m][r];
                }
                yy[m][0][0] = s0[0] + s0[1] + s0[2];
                yy[m][0][1] = s0[1] - s0[2] - s0[3];
                yy[m][1][0] = s1[0] + s1[1] + s1[2];
                yy[m][1][1] = s1[1] - s1[2] - s1[3];
            }
            if (EPI == 1) {
                const int ch = slice * 16 + 4 * g + r;
                const bool live = ch < Cout;
                int code;       // (bias - BatchNorm mean came in through the accumulator of position (1, 1): cst[])
                float vbest = pool_select_fast(yy[0][0][0], yy[1][0][0], yy[0][0][1], yy[1][0][1], yy[0][1][0], yy[1][1][0],
                                               yy[0][1][1], yy[1][1][1], code);
                if (bn_mean) vbest *= cst[32 + 4 * g + r];
                if (valid && live && th < Ho && tw < Wo) {
                    const size_t o = ((size_t)n * Cout + ch) * Ho * Wo + (size_t)th * Wo + tw;
                    y[o] = vbest;
                    idx[o] = (uint8_t)code;
                }
            } else if (EPI == 2) {
                // bias + max-feature-map [+ BatchNorm], no pool: the tile's 2x2 outputs and one byte with their 4
                // "second half won" bits (bit = 2 * row + col), (N, C, TH, TW)
                const int ch = slice * 16 + 4 * g + r;
                const bool live = ch < Cout;
                const float sc = cst[32 + 4 * g + r];
                float out[2][2];
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float va = yy[0][i][j], vb = yy[1][i][j];
                        const bool tb = mfm_takes_b(va, vb);
                        bits |= (uint32_t)tb << (2 * i + j);
                        out[i][j] = bn_mean ? (tb ? vb : va) * sc : (tb ? vb : va);
                    }
                if (valid && live) {
                    float *o = y + (((size_t)n * Cout + ch) * H + 2 * th) * W + 2 * tw;
                    const bool h1 = 2 * th + 1 < H, w1 = 2 * tw + 1 < W;
                    o[0] = out[0][0];
                    if (w1) o[1] = out[0][1];
                    if (h1) {
                        o[W] = out[1][0];
                        if (w1) o[W + 1] = out[1][1];
                    }
                    idx[((size_t)n * Cout + ch) * TH * TW + (size_t)th * TW + tw] = (uint8_t)bits;
                }
            } else if (EPI == 4 || EPI == 7) {
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const int ch = slice * 32 + (m + half) * 16 + 4 * g + r;
                    const bool live = ch < Cout;
                    if (EPI == 7) {
                        const float w0 = few_w[(m * 16 + 4 * g + r) * 2], w1 = few_w[(m * 16 + 4 * g + r) * 2 + 1];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            yy[m][e >> 1][e & 1] = fmaf(w1, xf[1][e >> 1][e & 1], fmaf(w0, xf[0][e >> 1][e & 1], yy[m][e >> 1][e & 1]));
                    }
                    float best = -INFINITY;
                    int code = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {      // ATen's scan: row-major, take when (v > best) || isnan(v)
                        const float v = yy[m][e >> 1][e & 1];
                        if (v > best || v != v) { best = v; code = e; }
                    }
                    if (valid && live && th < Ho && tw < Wo) {
                        const size_t o = ((size_t)n * Cout + ch) * Ho * Wo + (size_t)th * Wo + tw;
                        y[o] = best;
                        idx[o] = (uint8_t)code;
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const int ch = slice * 32 + (m + half) * 16 + 4 * g + r;
                    if (!(valid && ch < Cout)) continue;
                    if (EPI == 3) {
                        uint32_t bits = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = yy[m][e >> 1][e & 1];
                            bits |= (uint32_t)(v > 0.0f) << e;
                            yy[m][e >> 1][e & 1] = v > 0.0f ? v : v * ga.slope;
                        }
                        // the activation's sign bytes for EPI 6 (v > 0 and output > 0 are the same statement for slope > 0);
                        // positions outside an odd-sized plane read 0
                        if (idx) {
                            if (2 * tw + 1 >= W) bits &= 0x5u;
                            if (2 * th + 1 >= H) bits &= 0x3u;
                            idx[(((size_t)n * Cout + ch) * TH + th) * TW + tw] = (uint8_t)bits;
                        }
                    }
                    float *o = y + (((size_t)n * Cout + ch) * H + 2 * th) * W + 2 * tw;
                    const bool h1 = 2 * th + 1 < H;
                    if (EPI == 5) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) yy[m][e >> 1][e & 1] *= hv5[r][m][e >> 1][e & 1] > 0.0f ? 1.0f : ga.slope;
                    }
                    if (EPI == 6) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) yy[m][e >> 1][e & 1] *= (hb6[r][m] >> e) & 1u ? 1.0f : ga.slope;
                    }
                    if ((W & 1) == 0) {   // rows are 8-byte aligned: one 64-bit store per tile row
                        *reinterpret_cast<f32x2 *>(o) = (f32x2){yy[m][0][0], yy[m][0][1]};
                        if (h1) *reinterpret_cast<f32x2 *>(o + W) = (f32x2){yy[m][1][0], yy[m][1][1]};
                    } else {
                        const bool w1 = 2 * tw + 1 < W;
                        o[0] = yy[m][0][0];
                        if (w1) o[1] = yy[m][0][1];
                        if (h1) {
                            o[W] = yy[m][1][0];
                            if (w1) o[W + 1] = yy[m][1][1];
                        }
                    }
                }
            }
        }
    }
}

// d(conv out) (N, 2C, H, W) of the un-pooled block: gy * gscale goes to the half the tile byte names, 0 to the other.
__global__ __launch_bounds__(256) void wino_mfm_backward_kernel(const float *__restrict__ gy, const uint8_t *__restrict__ sel,
                                                                const float *__restrict__ gscale, float *__restrict__ gout,
                                                                int C, int H, int W, int64_t total) {
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), h = (int)((i / W) % H);
        const int64_t nc = i / ((int64_t)W * H);
        const int c = (int)(nc % C);
        const int64_t n = nc / C;
        const uint32_t bits = sel[(nc * TH + (h >> 1)) * TW + (w >> 1)];
        const bool tb = (bits >> (2 * (h & 1) + (w & 1))) & 1u;
        const float gv = gy[i] * (gscale ? gscale[c] : 1.0f);
        const int64_t o = ((n * 2 * C + c) * H + h) * W + w;
        gout[o] = tb ? 0.0f : gv;
        gout[o + (int64_t)C * H * W] = tb ? gv : 0.0f;
    }
}

// ADVSTEP_WINO_HALF_SLICE=0 (read at every call) keeps the single launch that multiplies the zero rows (A/B measurements).
inline bool half_slice_enabled() {
    const char *e = getenv("ADVSTEP_WINO_HALF_SLICE");
    return !(e && e[0] == '0');
}

template <int EPI, int SRC, bool GEN = false>
int launch_wino(const float *x, const uint8_t *xsel, const float *U, const float *bias, const float *bn_mean,
                const float *bn_invstd, float *y, uint8_t *idx, int64_t N, int64_t K, int64_t H, int64_t W, int64_t Cout,
                int slices, hipStream_t st, GenArgs ga = GenArgs{nullptr, 0, 0, 1.0f, 0, 0, 0}) {
    const int cus = 256;
    const int chunks = (int)ceil_div(K, kChunkCin);
    const bool stream = chunks > kMaxResident;
    const int64_t groups = ceil_div(N * ((H + 1) / 2) * ((W + 1) / 2), 16);
    const size_t lds = (size_t)(stream ? (SRC == 1 ? 4 : 2) : chunks) * kChunkFloats * sizeof(float);
    const bool wodd = SRC == 0 && (W & 1);
    // ADVSTEP_WINO_RANGE_MULT=k (read per call, default 1): k times the workgroups, each walking 1 / k of the tile groups - a
    // grid that is NOT persistent, so that workgroups of another stream's launch find compute units while this one runs (round 6
    // experiment; the weights are staged once per workgroup, i.e. k times as often)
    const char *em = getenv("ADVSTEP_WINO_RANGE_MULT");
    const int mult = em && em[0] >= '1' && em[0] <= '8' ? em[0] - '0' : 1;
    auto go = [&](auto kernel, int n_slices, int slice0) {
        int ranges = mult * cus / n_slices;
        if ((int64_t)ranges * kWaves > groups) ranges = (int)ceil_div(groups, kWaves);
        if (ranges < 1) ranges = 1;
        // the slices of one tile range read the same input tiles at the same time: put them on ONE XCD (workgroups b, b + 8, ...
        // share an L2) so all but one of the reads hit it: 2-3 % on L13 forward and SpecRNet's block2.
        // ADVSTEP_WINO_XCD=0 (read per call): round-robin slices (A/B)
        const char *e = getenv("ADVSTEP_WINO_XCD");
        ga.xcd = 0;
        // — only where rounding the ranges down to a multiple of 8 does not add a pass over the tile groups
        const int ranges8 = ranges & ~7;
        if (!(e && e[0] == '0') && n_slices > 1 && ranges8 >= 8 &&
            ceil_div(groups, (int64_t)ranges8 * kWaves) == ceil_div(groups, (int64_t)ranges * kWaves)) {
            ranges = ranges8;
            ga.xcd = 1;
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3((unsigned)(n_slices * ranges)), dim3(kThreads), lds, st, x, xsel, U, bias, bn_mean, bn_invstd,
                           y, idx, (int)N, (int)K, (int)H, (int)W, (int)Cout, n_slices, ranges, slice0, ga);
    };
    // plain-store epilogues with a half-empty last slice (Cout % 32 in 1..16): that slice on its own, one accumulator tile
    int full = slices;
    if constexpr (EPI == 0 || EPI == 3 || EPI == 5 || EPI == 6) {
        const int live_last = (int)(Cout - (int64_t)(slices - 1) * 32);
#ifndef WINO_TIMING_NT1          // timing-only builds (tools/build_variant.sh ... -DWINO_TIMING_NT1=1): the last slice ALWAYS as one
#define WINO_TIMING_NT1 0       // accumulator tile, i.e. rows 16..31 of a 17..32-row slice dropped (wrong results) - what SpecRNet's
#endif                          // 20-row layers would cost if their 12 dead rows were free
        if ((live_last <= 16 || WINO_TIMING_NT1) && half_slice_enabled()) {
            full = slices - 1;
            if (stream) wodd ? go(wino3x3_kernel<EPI, true, SRC, 1, GEN, SRC == 0>, 1, slices - 1) : go(wino3x3_kernel<EPI, true, SRC, 1, GEN>, 1, slices - 1);
            else wodd ? go(wino3x3_kernel<EPI, false, SRC, 1, GEN, SRC == 0>, 1, slices - 1) : go(wino3x3_kernel<EPI, false, SRC, 1, GEN>, 1, slices - 1);
        }
    }
    if constexpr (EPI == 0 && SRC == 1 && !GEN) {
        // a ONE-slice layer whose tile groups do not fill the chip's 2 048 wave slots even once: the slice as its two 16-row
        // halves, one accumulator tile each - twice the workgroups, half the matrix instructions per wave (the patch loads and
        // input transforms are done twice, on compute units that would have idled).  ADVSTEP_WINO_HALVES=0: one launch (A/B)
        const char *eh = getenv("ADVSTEP_WINO_HALVES");
        if (full == 1 && slices == 1 && Cout == 32 && groups <= (int64_t)cus * kWaves / 2 && !(eh && eh[0] == '0')) {
            ga.halves = 1;
            stream ? go(wino3x3_kernel<EPI, true, SRC, 1, GEN>, 2, 0) : go(wino3x3_kernel<EPI, false, SRC, 1, GEN>, 2, 0);
            return status_after_launch();
        }
    }
    if (full > 0) {
        if (stream) wodd ? go(wino3x3_kernel<EPI, true, SRC, 2, GEN, SRC == 0>, full, 0) : go(wino3x3_kernel<EPI, true, SRC, 2, GEN>, full, 0);
        else wodd ? go(wino3x3_kernel<EPI, false, SRC, 2, GEN, SRC == 0>, full, 0) : go(wino3x3_kernel<EPI, false, SRC, 2, GEN>, full, 0);
    }
    return status_after_launch();
}

}  // namespace

#define WINO_REQUIRE(cond) \
    do {                   \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

int advstep_conv3x3_supported(int64_t Cin, int64_t Cout) {
    return Cin >= 32 && Cin % 16 == 0 && Cout >= 32 && Cout % 32 == 0 && Cin <= 256 && Cout <= 256;
}

size_t advstep_conv3x3_prepared_floats(int64_t Cin, int64_t Cout, int mode) {
    if (!advstep_conv3x3_supported(Cin, Cout) || mode < 0 || mode > 2) return 0;
    const int64_t K = mode == 0 ? Cin : Cout;
    const int64_t slices = mode == 0 ? ceil_div(Cout / 2, 16) : ceil_div(Cin, 32);
    return (size_t)(slices * (K / kChunkCin) * kChunkFloats);
}

int advstep_conv3x3_prepare_f32(const float *weight, const float *gscale, float *U, int64_t Cin, int64_t Cout, int mode,
                                advstep_stream_t stream) {
    WINO_REQUIRE(weight && U && advstep_conv3x3_supported(Cin, Cout) && mode >= 0 && mode <= 2);
    WINO_REQUIRE(gscale == nullptr || mode != 0);
    const int K = (int)(mode == 0 ? Cin : Cout);
    const int slices = (int)(mode == 0 ? ceil_div(Cout / 2, 16) : ceil_div(Cin, 32));
    const int total = slices * 32 * K;
    hipLaunchKernelGGL(wino_prepare_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), weight,
                       gscale, U, (int)Cin, (int)Cout, mode, slices, K / kChunkCin);
    return status_after_launch();
}

int advstep_conv3x3_mfm_pool2_forward_f32(const float *x, const float *U, const float *bias, const float *bn_mean,
                                          const float *bn_invstd, float *y, uint8_t *idx, int64_t N, int64_t Cin, int64_t C,
                                          int64_t H, int64_t W, advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && advstep_conv3x3_supported(Cin, 2 * C));
    if (N == 0 || H / 2 == 0 || W / 2 == 0) return ADVSTEP_OK;
    WINO_REQUIRE(x && U && y && idx && (bn_mean == nullptr) == (bn_invstd == nullptr));
    WINO_REQUIRE((uint64_t)N * Cin * H * W * 4 < (1ull << 31) && (uint64_t)N * ((H + 1) / 2) * ((W + 1) / 2) < (1ull << 31));
    return launch_wino<1, 0>(x, nullptr, U, bias, bn_mean, bn_invstd, y, idx, N, Cin, H, W, C, (int)ceil_div(C, 16),
                             as_stream(stream));
}

size_t advstep_conv3x3_mfm_sel_bytes(int64_t N, int64_t C, int64_t H, int64_t W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)(N * C * ((H + 1) / 2) * ((W + 1) / 2));
}

int advstep_conv3x3_mfm_forward_f32(const float *x, const float *U, const float *bias, const float *bn_mean,
                                    const float *bn_invstd, float *y, uint8_t *sel, int64_t N, int64_t Cin, int64_t C,
                                    int64_t H, int64_t W, advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && advstep_conv3x3_supported(Cin, 2 * C));
    if (N == 0 || H == 0 || W == 0) return ADVSTEP_OK;
    WINO_REQUIRE(x && U && y && sel && (bn_mean == nullptr) == (bn_invstd == nullptr));
    WINO_REQUIRE((uint64_t)N * Cin * H * W * 4 < (1ull << 31) && (uint64_t)N * ((H + 1) / 2) * ((W + 1) / 2) < (1ull << 31));
    return launch_wino<2, 0>(x, nullptr, U, bias, bn_mean, bn_invstd, y, sel, N, Cin, H, W, C, (int)ceil_div(C, 16),
                             as_stream(stream));
}

int advstep_conv3x3_mfm_backward_f32(const float *gy, const uint8_t *sel, const float *gscale, float *gout, int64_t N,
                                     int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && C >= 0 && H >= 0 && W >= 0);
    const int64_t total = N * C * H * W;
    if (total == 0) return ADVSTEP_OK;
    WINO_REQUIRE(gy && sel && gout);
    const int64_t blocks = ceil_div(total, 256);
    hipLaunchKernelGGL(wino_mfm_backward_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       as_stream(stream), gy, sel, gscale, gout, (int)C, (int)H, (int)W, total);
    return status_after_launch();
}

int advstep_conv3x3_backward_data_f32(const float *gout, const float *U, float *gx, int64_t N, int64_t Cin, int64_t Cout,
                                      int64_t H, int64_t W, advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && advstep_conv3x3_supported(Cin, Cout));
    if (N == 0 || H == 0 || W == 0) return ADVSTEP_OK;
    WINO_REQUIRE(gout && U && gx);
    WINO_REQUIRE((uint64_t)N * Cout * H * W * 4 < (1ull << 31) && (uint64_t)N * ((H + 1) / 2) * ((W + 1) / 2) < (1ull << 31));
    return launch_wino<0, 0>(gout, nullptr, U, nullptr, nullptr, nullptr, gx, nullptr, N, Cout, H, W, Cin,
                             (int)ceil_div(Cin, 32), as_stream(stream));
}

int advstep_conv3x3_mfm_pool2_backward_f32(const float *gy, const uint8_t *idx, const float *U, float *gx, int64_t N,
                                           int64_t Cin, int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && advstep_conv3x3_supported(Cin, 2 * C));
    if (N == 0 || H == 0 || W == 0) return ADVSTEP_OK;
    WINO_REQUIRE(gx);
    if (H / 2 == 0 || W / 2 == 0)
        return hipMemsetAsync(gx, 0, (size_t)N * Cin * H * W * sizeof(float), as_stream(stream)) == hipSuccess ? ADVSTEP_OK
                                                                                                               : ADVSTEP_ELAUNCH;
    WINO_REQUIRE(gy && idx && U);
    WINO_REQUIRE((uint64_t)N * C * (H / 2) * (W / 2) < (1ull << 29) && (uint64_t)N * ((H + 1) / 2) * ((W + 1) / 2) < (1ull << 31));
    return launch_wino<0, 1>(gy, idx, U, nullptr, nullptr, nullptr, gx, nullptr, N, 2 * C, H, W, Cin, (int)ceil_div(Cin, 32),
                             as_stream(stream));
}

// ---- plain 3x3 convolutions of the detectors' residual blocks (include/advstep_detector.h) -----------------------------

int advstep_resconv_supported(int64_t K1, int64_t K2, int64_t rows) {
    return K1 >= 1 && K2 >= 0 && (K2 == 0 || K1 % 4 == 0) && K1 + K2 <= 256 && rows >= 1 && rows <= 256;
}

size_t advstep_resconv_prepared_floats(int64_t K1, int64_t K2, int64_t rows) {
    if (!advstep_resconv_supported(K1, K2, rows)) return 0;
    return (size_t)(ceil_div(rows, 32) * ceil_div(K1 + K2, kChunkCin) * kChunkFloats);
}

int advstep_resconv_prepare_f32(const float *w3, const float *w1, const float *rscale, const float *kscale, float *U,
                                int64_t rows, int64_t K1, int64_t K2, int transpose, advstep_stream_t stream) {
    WINO_REQUIRE(w3 && U && advstep_resconv_supported(K1, K2, rows) && (K2 == 0 || w1) && (transpose == 0 || transpose == 1));
    const int slices = (int)ceil_div(rows, 32), chunks = (int)ceil_div(K1 + K2, kChunkCin);
    const int total = slices * 32 * chunks * kChunkCin;
    hipLaunchKernelGGL(wino_prepare_plain_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w3, w1,
                       rscale, kscale, U, (int)rows, (int)K1, (int)K2, transpose, slices, chunks);
    return status_after_launch();
}

static int resconv_check(const float *x1, const float *x2, const float *U, const void *y, int64_t N, int64_t K1, int64_t K2,
                         int64_t rows, int64_t H, int64_t W) {
    WINO_REQUIRE(x1 && U && y && (K2 == 0 || x2));
    WINO_REQUIRE((uint64_t)N * (K1 > K2 ? K1 : K2) * H * W * 4 < (1ull << 31) && (uint64_t)N * rows * H * W * 4 < (1ull << 33) &&
                 (uint64_t)N * ((H + 1) / 2) * ((W + 1) / 2) < (1ull << 31));
    return ADVSTEP_OK;
}

int advstep_resconv_forward_act_f32(const float *x1, const float *x2, const float *U, const float *shift, float slope, float *y,
                                    uint8_t *act, int64_t N, int64_t K1, int64_t K2, int64_t rows, int64_t H, int64_t W,
                                    advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && advstep_resconv_supported(K1, K2, rows));
    if (N == 0 || H == 0 || W == 0) return ADVSTEP_OK;
    if (const int st = resconv_check(x1, x2, U, y, N, K1, K2, rows, H, W)) return st;
    const int64_t K = (K1 + K2 <= 8) ? 8 : ceil_div(K1 + K2, 4) * 4;      // at least two k-steps; an odd count is fine
    return launch_wino<3, 0, true>(x1, nullptr, U, shift, nullptr, nullptr, y, act, N, K, H, W, rows, (int)ceil_div(rows, 32),
                                   as_stream(stream), GenArgs{x2, (int)K1, (int)(K1 + K2), slope, 0, 0, 0});
}

int advstep_resconv_forward_f32(const float *x1, const float *x2, const float *U, const float *shift, float slope, float *y,
                                int64_t N, int64_t K1, int64_t K2, int64_t rows, int64_t H, int64_t W,
                                advstep_stream_t stream) {
    return advstep_resconv_forward_act_f32(x1, x2, U, shift, slope, y, nullptr, N, K1, K2, rows, H, W, stream);
}

int advstep_resconv_pool2_forward_f32(const float *x1, const float *x2, const float *U, const float *bias, float *y,
                                      uint8_t *sel, int64_t N, int64_t K1, int64_t K2, int64_t rows, int64_t H, int64_t W,
                                      advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && advstep_resconv_supported(K1, K2, rows));
    if (N == 0 || H / 2 == 0 || W / 2 == 0) return ADVSTEP_OK;
    WINO_REQUIRE(sel);
    if (const int st = resconv_check(x1, x2, U, y, N, K1, K2, rows, H, W)) return st;
    const int64_t K = (K1 + K2 <= 8) ? 8 : ceil_div(K1 + K2, 4) * 4;      // at least two k-steps; an odd count is fine
    return launch_wino<4, 0, true>(x1, nullptr, U, bias, nullptr, nullptr, y, sel, N, K, H, W, rows, (int)ceil_div(rows, 32),
                                   as_stream(stream), GenArgs{x2, (int)K1, (int)(K1 + K2), 1.0f, 0, 0, 0});
}

static int pooled_grad(const float *gy, const uint8_t *sel, const float *U, const float *h, const uint8_t *act, float slope,
                       float *g, int64_t N, int64_t K, int64_t rows, int64_t H, int64_t W, advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && advstep_resconv_supported(K, 0, rows));
    if (N == 0 || H == 0 || W == 0) return ADVSTEP_OK;
    WINO_REQUIRE(g);
    if (H / 2 == 0 || W / 2 == 0)
        return hipMemsetAsync(g, 0, (size_t)N * rows * H * W * sizeof(float), as_stream(stream)) == hipSuccess ? ADVSTEP_OK
                                                                                                                : ADVSTEP_ELAUNCH;
    WINO_REQUIRE(gy && sel && U);
    WINO_REQUIRE((uint64_t)N * K * (H / 2) * (W / 2) < (1ull << 29) && (uint64_t)N * rows * H * W * 4 < (1ull << 33) &&
                 (uint64_t)N * ((H + 1) / 2) * ((W + 1) / 2) < (1ull << 31));
    const int64_t Kp = K <= 8 ? 8 : ceil_div(K, 4) * 4;
    const GenArgs ga{nullptr, (int)K, (int)K, slope, 0, 0, 0};
    if (act)
        return launch_wino<6, 2, true>(gy, sel, U, nullptr, nullptr, nullptr, g, const_cast<uint8_t *>(act), N, Kp, H, W, rows,
                                       (int)ceil_div(rows, 32), as_stream(stream), ga);
    if (h)
        return launch_wino<5, 2, true>(gy, sel, U, nullptr, h, nullptr, g, nullptr, N, Kp, H, W, rows, (int)ceil_div(rows, 32),
                                       as_stream(stream), ga);
    return launch_wino<0, 2, true>(gy, sel, U, nullptr, nullptr, nullptr, g, nullptr, N, Kp, H, W, rows, (int)ceil_div(rows, 32),
                                   as_stream(stream), ga);
}

int advstep_resconv_pool2_forward_few_f32(const float *x1, const float *x2, const float *U, const float *wd, const float *bias,
                                          float *y, uint8_t *sel, int64_t N, int64_t K1, int64_t K2, int64_t rows, int64_t H,
                                          int64_t W, advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && (K2 == 1 || K2 == 2) && advstep_resconv_supported(K1, 0, rows));
    if (N == 0 || H / 2 == 0 || W / 2 == 0) return ADVSTEP_OK;
    WINO_REQUIRE(sel && x2 && wd);
    if (const int st = resconv_check(x1, nullptr, U, y, N, K1, 0, rows, H, W)) return st;
    const int64_t K = K1 <= 8 ? 8 : ceil_div(K1, 4) * 4;
    return launch_wino<7, 0, true>(x1, nullptr, U, bias, wd, nullptr, y, sel, N, K, H, W, rows, (int)ceil_div(rows, 32),
                                   as_stream(stream), GenArgs{x2, (int)K1, (int)K1, 1.0f, 0, (int)K2, 0});
}

int advstep_resconv_pooled_grad_f32(const float *gy, const uint8_t *sel, const float *U, const float *h, float slope, float *g,
                                    int64_t N, int64_t K, int64_t rows, int64_t H, int64_t W, advstep_stream_t stream) {
    return pooled_grad(gy, sel, U, h, nullptr, slope, g, N, K, rows, H, W, stream);
}

int advstep_resconv_pooled_grad_act_f32(const float *gy, const uint8_t *sel, const float *U, const uint8_t *act, float slope,
                                        float *g, int64_t N, int64_t K, int64_t rows, int64_t H, int64_t W,
                                        advstep_stream_t stream) {
    if (!act) return ADVSTEP_EINVAL;
    return pooled_grad(gy, sel, U, nullptr, act, slope, g, N, K, rows, H, W, stream);
}

}  // extern "C"
