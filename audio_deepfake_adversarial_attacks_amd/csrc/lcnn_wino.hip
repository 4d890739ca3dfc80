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

// ---- weight transform: U = G g G^T into the chunked layout ------------------------------------------------------------
// mode 0 (forward, max-feature-map pairs): output row (slice, j, m) = conv channel m * C + slice * 16 + j, g = weight.
// mode 1 (input gradient): the convolution that maps d(conv out) (2C channels) to d(conv in) (Cin channels) has kernel
//         g'[ci][co][a][b] = weight[co][ci][2 - a][2 - b]; output row (slice, j, m) = input channel slice*32 + m*16 + j.
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
        const int ci = slice * 32 + m * 16 + j;
        live = ci < Cin;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = live ? weight[((int64_t)k * Cin + ci) * 9 + (2 - a) * 3 + (2 - b)] : 0.0f;
        // a per-channel factor on d(conv out) — the folded BatchNorm's invstd of max-feature-map channel k % C — is a
        // factor on row k of the operand
        if (kscale) {
            const float f = kscale[k % (Cout / 2)];
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

// ---- the convolution ---------------------------------------------------------------------------------------------------
// EPI 0: plain store of min(32, Cout - slice * 32) channels per slice (the input-gradient convolution).
// EPI 1: bias + max-feature-map + 2x2 pool [+ BatchNorm]; Cout = number of max-feature-map channels C.
// EPI 2: bias + max-feature-map [+ BatchNorm] without the pool; one selection byte per 2x2 tile.
// STREAM: K > 64, U chunks double-buffered through LDS (one barrier per chunk); otherwise all chunks stay resident.
// grid = slices * ranges workgroups; workgroup b: slice b % slices, tile range b / slices.
// SRC 0: the input is a dense tensor x (N, K, H, W).
// SRC 1: the input is d(conv out) of a max-feature-map + 2x2 pool block, given in its compact form — the pooled gradient
//        gy (N, C, H/2, W/2) and the selection bytes (advstep_mfm_pool2_forward_f32's encoding), K = 2C: channel k of
//        half k / C at conv position (h, w) carries gy[k % C][h/2][w/2] if that position of that half won, else 0.  The
//        4x4 patch of a lane is expanded from the 3x3 pooled cells around its tile; the dense gradient never exists.
template <int EPI, bool STREAM, int SRC>
__global__ __launch_bounds__(kThreads) void wino3x3_kernel(const float *__restrict__ x, const uint8_t *__restrict__ xsel,
                                                           const float *__restrict__ U,
                                                           const float *__restrict__ bias,
                                                           const float *__restrict__ bn_mean,
                                                           const float *__restrict__ bn_invstd, float *__restrict__ y,
                                                           uint8_t *__restrict__ idx, int N, int K, int H, int W, int Cout,
                                                           int slices, int ranges) {
    extern __shared__ __attribute__((aligned(16))) float u_s[];
    const int slice = blockIdx.x % slices, range = blockIdx.x / slices;
    const int chunks = K / kChunkCin, steps = K / 4;
    const float *Usl = U + (int64_t)slice * chunks * kChunkFloats;
    auto copy_chunk = [&](int chunk, int buf) {
        const float4 *src = reinterpret_cast<const float4 *>(Usl + (int64_t)chunk * kChunkFloats);
        float4 *dst = reinterpret_cast<float4 *>(u_s + buf * kChunkFloats);
#pragma unroll
        for (int i = 0; i < kChunkFloats / 4 / kThreads; ++i) dst[threadIdx.x + i * kThreads] = src[threadIdx.x + i * kThreads];
    };
    if (!STREAM) {
        for (int c = 0; c < chunks; ++c) copy_chunk(c, c);
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, nl = lane & 15;
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    const int tiles = N * TH * TW, groups = (tiles + 15) >> 4;
    const int iters = (groups + ranges * kWaves - 1) / (ranges * kWaves);   // the same for every workgroup
    const uint32_t plane = (uint32_t)(H * W);
    // raw buffer over x: an out-of-range offset reads as 0 — the convolution's zero padding, for free
    const int Hs = H >> 1, Ws = W >> 1, Cs = K >> 1;      // SRC 1: pooled grid, max-feature-map channels
    const uint32_t cplane = (uint32_t)(Hs * Ws);
    const size_t src_elems = SRC == 0 ? (size_t)N * K * plane : (size_t)N * Cs * cplane;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)(src_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(SRC == 1 ? xsel : reinterpret_cast<const uint8_t *>(x)), 0,
                                          (int)src_elems, 0x00020000);

    for (int it = 0; it < iters; ++it) {
        const int grp = (it * ranges + range) * kWaves + wave;
        const int t = grp * 16 + nl;
        const bool valid = grp < groups && t < tiles;
        const int tt = valid ? t : 0;
        const int n = tt / (TH * TW), rem = tt - n * (TH * TW), th = rem / TW, tw = rem - th * TW;
        // patch addressing: one lane base + compile-time (p, q) strides; the 16 validity flags live in scalar registers
        // as lane masks, and an invalid tap gets an out-of-range offset (reads 0) when the load is issued
        const uint32_t lane_base = (((uint32_t)(n * K + g)) * plane + (uint32_t)((2 * th - 1) * W + (2 * tw - 1))) * 4u;
        bool ok[4][4];
        uint32_t cell[3][3];     // SRC 1: element offsets of the 3x3 pooled cells around the tile (0x20000000 = outside)
        if (SRC == 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int hh = 2 * th - 1 + p, ww = 2 * tw - 1 + q;
                    ok[p][q] = valid && hh >= 0 && hh < H && ww >= 0 && ww < W;
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
        f32x4 acc[16][2];

        // a patch in flight: SRC 0 the 16 taps; SRC 1 the 9 pooled gradients + their 9 selection bytes
        struct Patch {
            float v[SRC == 0 ? 16 : 9];
            uint32_t code[SRC == 0 ? 1 : 9];
        };
        auto load_patch = [&](Patch &dst, int s) {
            if (SRC == 0) {
                const uint32_t soff = (uint32_t)(4 * s) * plane * 4u;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t vo = ok[p][q] ? lane_base + (uint32_t)((p * W + q) * 4) : 0x80000000u;
                        dst.v[p * 4 + q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vo, soff, 0));
                    }
            } else {
                const int k0 = 4 * s, c0 = k0 >= Cs ? k0 - Cs : k0;      // wave-uniform: channel of lane group 0
                const uint32_t soff = (uint32_t)c0 * cplane;
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const uint32_t e = cell[i / 3][i % 3];
                    dst.v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, e << 2, soff << 2, 0));
                    dst.code[i] = __builtin_amdgcn_raw_buffer_load_b8(sr, e, soff, 0);
                }
            }
        };
        // the 4x4 taps of a patch: SRC 1 routes each pooled gradient to the one conv position (of the one half) that won
        auto taps = [&](const Patch &src, int s, float (&d)[4][4]) {
            if (SRC == 0) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[p][q] = src.v[p * 4 + q];
            } else {
                const uint32_t half_bit = 4 * s >= Cs ? 4u : 0u;         // wave-uniform
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
                a[grp][0] = *reinterpret_cast<const f32x2 *>(us + (2 * grp) * (kChunkCin * 32));
                a[grp][1] = *reinterpret_cast<const f32x2 *>(us + (2 * grp + 1) * (kChunkCin * 32));
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
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                if (grp + 2 < 8) request(grp + 2);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int xi = 2 * grp + e;
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[grp][e].x, v[xi >> 2][xi & 3], decltype(first)::value ? zero : acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[grp][e].y, v[xi >> 2][xi & 3], decltype(first)::value ? zero : acc[xi][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // lane's A address inside a chunk buffer for k-step s: cin_in_chunk = (s & 3) * 4 + g
        auto a_ptr = [&](int s, int buf) { return u_s + buf * kChunkFloats + (((s & 3) * 4 + g) * 16 + nl) * 2; };

        Patch da, db;
        if (STREAM) {
            __syncthreads();            // previous iteration's readers are done with both buffers
            copy_chunk(0, 0);
            __syncthreads();
        }
        load_patch(da, 0);
        load_patch(db, 1);
        step(da, 0, a_ptr(0, 0), std::true_type{});
        load_patch(da, 2);
        step(db, 1, a_ptr(1, 0), std::false_type{});
        if (STREAM) copy_chunk(1, 1);   // chunks >= 2 always here; lands while chunk 0's last steps run
#pragma unroll 1
        for (int s = 2; s < steps; s += 2) {
            if (STREAM && (s & 3) == 0) {
                __syncthreads();        // chunk s/4 is complete in its buffer; chunk s/4 - 1 is free
                if (s / 4 + 1 < chunks) copy_chunk(s / 4 + 1, (s / 4 + 1) & 1);
            }
            const int buf = STREAM ? ((s >> 2) & 1) : (s >> 2);
            load_patch(db, s + 1);      // in flight while this step's matrix instructions run
            step(da, s, a_ptr(s, buf), std::false_type{});
            if (s + 2 < steps) load_patch(da, s + 2);
            step(db, s + 1, a_ptr(s + 1, buf), std::false_type{});
        }

        // epilogue: Y = A^T M A per (channel, tile)
        const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float yy[2][2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
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
                const int chs = live ? ch : 0;
                const float ba = bias ? bias[chs] : 0.0f, bb = bias ? bias[chs + Cout] : 0.0f;
                int code;
                float vbest = pool_select(yy[0][0][0] + ba, yy[1][0][0] + bb, yy[0][0][1] + ba, yy[1][0][1] + bb,
                                          yy[0][1][0] + ba, yy[1][1][0] + bb, yy[0][1][1] + ba, yy[1][1][1] + bb, code);
                if (bn_mean) vbest = (vbest - bn_mean[chs]) * bn_invstd[chs];
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
                const int chs = live ? ch : 0;
                const float ba = bias ? bias[chs] : 0.0f, bb = bias ? bias[chs + Cout] : 0.0f;
                const float mu = bn_mean ? bn_mean[chs] : 0.0f, sc = bn_mean ? bn_invstd[chs] : 1.0f;
                float out[2][2];
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float va = yy[0][i][j] + ba, vb = yy[1][i][j] + bb;
                        const bool tb = mfm_takes_b(va, vb);
                        bits |= (uint32_t)tb << (2 * i + j);
                        out[i][j] = ((tb ? vb : va) - mu) * sc;
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
            } else {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int ch = slice * 32 + m * 16 + 4 * g + r;
                    if (!(valid && ch < Cout)) continue;
                    float *o = y + (((size_t)n * Cout + ch) * H + 2 * th) * W + 2 * tw;
                    const bool h1 = 2 * th + 1 < H;
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


// ---- the position-split variant: two waves share a tile group, 8 Winograd positions each -------------------------------------
// Measured on the kernel above (profiles/r01_model_kernel_pmc_wait.txt, profiles/r02_wino_notes.txt): per matrix instruction
// the waves issue about three other VALU instructions — the input transform B^T d B and the tap addresses of a k-step serve
// only 32 output channels — and fp32 matrix time and VALU time ADD on this part; a wave issues one
// VALU instruction per ~4.4 cycles, two waves of a SIMD together one per ~2.2, so giving a wave the whole register file
// (one wave per SIMD, 48-64 channels per wave: built, measured 466 us against 385 us on the largest layer) loses more on
// everything outside the k loop than it wins inside.  This variant keeps two waves per SIMD and widens differently:
//   * the two waves of a PAIR work on the same 16 tiles; half h owns rows 2h, 2h+1 of the 4x4 Winograd positions.  Row a of
//     V = B^T d B needs only row a of B^T d, so a half needs 3 of the 4 patch rows (12 taps) and 16 of the 32 transform adds:
//     nothing is computed twice, and its 8 positions x MT accumulator tiles (MT = 2, 3, 4: 32 / 48 / 64 output rows) fit the
//     register budget of the 32-row kernel — the transform and the taps of a k-step are amortised over up to twice the rows;
//   * tap offsets are hoisted out of the k loop, patches are requested a whole half-chunk (two k-steps) ahead, and no load
//     sits inside a branch (at a join the compiler waits for vmcnt(0), i.e. for the prefetch it has just issued);
//   * U streams through LDS as [xi][cin][16 j][MT] chunks of CC input channels; a streamed chunk is loaded into registers at
//     the head of the previous chunk's k-steps and written after them (no load latency in front of the barrier);
//   * epilogue: each half forms its share of Y = A^T M A (rows of M it owns), the halves swap the shares of the accumulator
//     registers r they do not finalise through LDS (half 0 finalises r = 0, 1, half 1 r = 2, 3) and add; bias, max-feature-map,
//     pool, BatchNorm and the stores then run on both halves, on disjoint channels;
//   * forward pairs: tiles [0, MT/2) hold the first halves of 16 max-feature-map pairs each, tiles [MT/2, 2 (MT/2)) the second
//     halves (same lane, same slot); an odd MT adds a mixed tile with 8 pairs — rows j < 8 first halves, rows j >= 8 second
//     halves, i.e. lanes l and l ^ 32, joined by one cross-lane exchange.
constexpr int kSplitWaves = 8, kSplitThreads = kSplitWaves * 64, kSplitPairs = 4;

__host__ __device__ constexpr int split_chunk_cin(int MT) { return MT == 4 ? 8 : 16; }
__host__ __device__ constexpr int split_chunk_floats(int MT) { return 16 * split_chunk_cin(MT) * 16 * MT; }
__host__ __device__ constexpr int split_pairs_per_slice(int MT) { return 16 * (MT / 2) + 8 * (MT & 1); }
__host__ __device__ constexpr int split_exchange_floats(int MT) { return 8 * MT * kSplitThreads; }
// chunks that may stay resident in LDS next to the exchange area (160 KiB per CU)
__host__ __device__ constexpr int split_max_resident(int MT) {
    return (160 * 1024 - split_exchange_floats(MT) * 4) / (split_chunk_floats(MT) * 4);
}

// U: [slice][chunk][xi][chunk cin][16 j][MT], zero where a row / channel does not exist.
__global__ void wino_prepare_split_kernel(const float *__restrict__ weight, const float *__restrict__ kscale,
                                          float *__restrict__ U, int Cin, int Cout, int mode, int slices, int chunks, int MT) {
    const int K = mode == 0 ? Cin : Cout;
    const int rows = 16 * MT, total = slices * rows * K;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = i % K, row = i / K, slice = row / rows, m = (row % rows) / 16, j = row % 16;
    const int CC = split_chunk_cin(MT);
    float g[3][3];
    bool live;
    if (mode == 0) {
        const int C = Cout / 2, half_tiles = MT / 2;
        int p, half;
        if (m < half_tiles) { p = m * 16 + j; half = 0; }
        else if (m < 2 * half_tiles) { p = (m - half_tiles) * 16 + j; half = 1; }
        else { p = half_tiles * 16 + (j & 7); half = j >> 3; }
        p += slice * split_pairs_per_slice(MT);
        live = p < C;
        const int co = half * C + p;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = live ? weight[((int64_t)co * Cin + k) * 9 + a * 3 + b] : 0.0f;
    } else {
        const int ci = slice * rows + m * 16 + j;
        live = ci < Cin;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = live ? weight[((int64_t)k * Cin + ci) * 9 + (2 - a) * 3 + (2 - b)] : 0.0f;
        if (kscale) {
            const float f = kscale[k % (Cout / 2)];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] *= f;
        }
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
    const int chunk = k / CC, kc = k % CC;
    float *dst = U + ((int64_t)(slice * chunks + chunk)) * split_chunk_floats(MT) + (kc * 16 + j) * MT + m;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) dst[xi * (CC * 16 * MT)] = u[xi >> 2][xi & 3];
}

template <int MT> struct SplitA;
template <> struct SplitA<2> { typedef f32x2 type; };
template <> struct SplitA<3> { typedef float type __attribute__((ext_vector_type(3))); };
template <> struct SplitA<4> { typedef f32x4 type; };

// Everything a half needs, as a struct of references captured once (the k loop below is instantiated per half).
template <int EPI, bool STREAM, int SRC, int MT>
struct SplitCtx {
    const float *Usl;
    float *u_s, *x_s;           // chunk buffers; exchange area
    __amdgpu_buffer_rsrc_t xr, sr;
    const float *bias, *bn_mean, *bn_invstd;
    float *y;
    uint8_t *idx;
    int N, K, H, W, Cout, slice, range, ranges, chunks, steps, pair, lane, g, nl, TH, TW, tiles, groups, iters, Hs, Ws, Cs;
    uint32_t plane, cplane;
};

template <int EPI, bool STREAM, int SRC, int MT, int HALF>
__device__ __forceinline__ void split_run(const SplitCtx<EPI, STREAM, SRC, MT> &c) {
    constexpr int CC = split_chunk_cin(MT), SPC = CC / 4, CF = split_chunk_floats(MT);
    constexpr int XI_STRIDE = CC * 16 * MT;
    constexpr int kCopy = CF / 4 / kSplitThreads;            // float4 per thread per chunk: 4 (MT 2, 4) or 6 (MT 3)
    typedef typename SplitA<MT>::type avec;
    const int K = c.K, H = c.H, W = c.W, Cout = c.Cout, g = c.g, nl = c.nl, TH = c.TH, TW = c.TW;
    const int Hs = c.Hs, Ws = c.Ws, Cs = c.Cs, chunks = c.chunks, steps = c.steps;
    const uint32_t plane = c.plane, cplane = c.cplane;
    float *u_s = c.u_s;
    const int tid = threadIdx.x;

    auto chunk_load = [&](int chunk, f32x4 (&r)[kCopy]) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(c.Usl + (int64_t)chunk * CF);
#pragma unroll
        for (int i = 0; i < kCopy; ++i) r[i] = src[tid + i * kSplitThreads];
    };
    auto chunk_store = [&](int buf, const f32x4 (&r)[kCopy]) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(u_s + buf * CF);
#pragma unroll
        for (int i = 0; i < kCopy; ++i) dst[tid + i * kSplitThreads] = r[i];
    };
    {
        f32x4 r[kCopy];
        for (int cc = 0; cc < (STREAM ? 1 : chunks); ++cc) {
            chunk_load(cc, r);
            chunk_store(cc, r);
        }
        __syncthreads();
    }
    int ring = 0;       // STREAM: chunks consumed so far; the chunk in use sits in buffer ring & 1

    // taps of this half: patch rows HALF .. HALF + 2.  SRC 0: 12 byte offsets (out-of-image taps and invalid tiles point outside
    // the buffer and read 0); SRC 1: the pooled cells of rows (HALF + 1) >> 1 .. — cell rows HALF, HALF + 1 — 6 element offsets
    constexpr int NOFF = SRC == 0 ? 12 : 6;
    struct Patch {
        float v[NOFF];
        uint32_t code[SRC == 0 ? 1 : 6];
    };
    for (int it = 0; it < c.iters; ++it) {
        const int grp = (it * c.ranges + c.range) * kSplitPairs + c.pair;
        const int t = grp * 16 + nl;
        const bool valid = grp < c.groups && t < c.tiles;
        const int tt = valid ? t : 0;
        const int n = tt / (TH * TW), rem = tt - n * (TH * TW), th = rem / TW, tw = rem - th * TW;
        uint32_t off[NOFF];
        if (SRC == 0) {
            const uint32_t lane_base = (((uint32_t)(n * K + g)) * plane + (uint32_t)((2 * th - 1) * W + (2 * tw - 1))) * 4u;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int hh = 2 * th - 1 + HALF + p, ww = 2 * tw - 1 + q;
                    const bool ok = valid && hh >= 0 && hh < H && ww >= 0 && ww < W;
                    off[p * 4 + q] = ok ? lane_base + (uint32_t)(((HALF + p) * W + q) * 4) : 0x80000000u;
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ci = th - 1 + HALF + i, cj = tw - 1 + j;
                    const bool in = valid && ci >= 0 && ci < Hs && cj >= 0 && cj < Ws;
                    off[i * 3 + j] = in ? ((uint32_t)(n * Cs + g)) * cplane + (uint32_t)(ci * Ws + cj) : 0x20000000u;
                }
        }
        auto load_patch = [&](Patch &dst, int s) {
            if (SRC == 0) {
                const uint32_t soff = (uint32_t)(4 * s) * plane * 4u;
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    dst.v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.xr, off[i], soff, 0));
            } else {
                const int k0 = 4 * s, c0 = k0 >= Cs ? k0 - Cs : k0;      // wave-uniform: channel of lane group 0
                const uint32_t soff = (uint32_t)c0 * cplane;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    dst.v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.xr, off[i] << 2, soff << 2, 0));
                    dst.code[i] = __builtin_amdgcn_raw_buffer_load_b8(c.sr, off[i], soff, 0);
                }
            }
        };
        // patch rows HALF .. HALF + 2 as d[0..2][q]
        auto taps = [&](const Patch &src, int s, float (&d)[3][4]) {
            if (SRC == 0) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[p][q] = src.v[p * 4 + q];
            } else {
                const uint32_t half_bit = 4 * s >= Cs ? 4u : 0u;         // wave-uniform
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pr = HALF + p;                         // patch row 0..3
                        const int i = (((pr + 1) >> 1) - HALF) * 3 + ((q + 1) >> 1);
                        const uint32_t want = half_bit | (uint32_t)((((pr + 1) & 1) << 1) | ((q + 1) & 1));
                        d[p][q] = src.code[i] == want ? src.v[i] : 0.0f;
                    }
            }
        };
        f32x4 acc[8][MT];
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        // one k-step of this half: positions (2 HALF + a, b), a = 0, 1.  B^T d rows: r0 = d0 - d2, r1 = d1 + d2, r2 = d2 - d1,
        // r3 = d1 - d3; with e = patch rows HALF .. HALF + 2:  half 0: (e0 - e2, e1 + e2), half 1: (e1 - e0, e0 - e2)
        auto step = [&](const Patch &patch, int s_idx, const float *us, auto first) {
            avec a[4][2];
            auto request = [&](int grp4) {
                a[grp4][0] = *reinterpret_cast<const avec *>(us + (2 * grp4) * XI_STRIDE);
                a[grp4][1] = *reinterpret_cast<const avec *>(us + (2 * grp4 + 1) * XI_STRIDE);
            };
            request(0);
            request(1);
            __builtin_amdgcn_sched_barrier(0);
            float e[3][4], tr[2][4], v[2][4];
            taps(patch, s_idx, e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dif = e[0][q] - e[2][q];
                tr[0][q] = HALF == 0 ? dif : e[1][q] - e[0][q];
                tr[1][q] = HALF == 0 ? e[1][q] + e[2][q] : dif;
            }
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
                v[a2][0] = tr[a2][0] - tr[a2][2];
                v[a2][1] = tr[a2][1] + tr[a2][2];
                v[a2][2] = tr[a2][2] - tr[a2][1];
                v[a2][3] = tr[a2][1] - tr[a2][3];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int grp4 = 0; grp4 < 4; ++grp4) {
                if (grp4 + 2 < 4) request(grp4 + 2);
#pragma unroll
                for (int ee = 0; ee < 2; ++ee) {
                    const int xl = 2 * grp4 + ee;                        // local position 0..7 = (a, b)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[xl][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[grp4][ee][m], v[xl >> 2][xl & 3],
                                                                          decltype(first)::value ? zero : acc[xl][m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // lane's A address inside a chunk buffer for k-step s: position 8 HALF, cin_in_chunk = (s % SPC) * 4 + g
        auto a_ptr = [&](int s, int buf) {
            return u_s + buf * CF + (8 * HALF) * XI_STRIDE + (((s % SPC) * 4 + g) * 16 + nl) * MT;
        };

        Patch a0, b0, a1, b1;
        load_patch(a0, 0);
        load_patch(b0, 1);
        f32x4 nxt[kCopy];
        // (pos = position of a k-step inside the 4-step unrolled body: the chunk phase is known at compile time)
        auto chunk_head = [&](int s, auto pos) {
            if (STREAM && decltype(pos)::value % SPC == 0) {
                const int cn = s / SPC + 1;
                chunk_load(cn == chunks ? 0 : cn, nxt);
            }
        };
        auto chunk_tail = [&](auto pos) {
            if (STREAM && decltype(pos)::value % SPC == SPC - 1) {
                chunk_store((ring + 1) & 1, nxt);
                ++ring;
                __syncthreads();
            }
        };
        typedef std::integral_constant<int, 0> P0;
        typedef std::integral_constant<int, 1> P1;
        typedef std::integral_constant<int, 2> P2;
        typedef std::integral_constant<int, 3> P3;
        auto buf_of = [&](int s) { return STREAM ? (ring & 1) : s / SPC; };
        const int last = steps - 1;
        // first four k-steps (the first one initialises the accumulators), then the rest
        chunk_head(0, P0{});
        load_patch(a1, 2);
        load_patch(b1, 3);
        step(a0, 0, a_ptr(0, buf_of(0)), std::true_type{});
        chunk_tail(P0{});
        chunk_head(1, P1{});
        step(b0, 1, a_ptr(1, buf_of(1)), std::false_type{});
        chunk_tail(P1{});
        chunk_head(2, P2{});
        load_patch(a0, 4);
        load_patch(b0, 5);
        step(a1, 2, a_ptr(2, buf_of(2)), std::false_type{});
        chunk_tail(P2{});
        chunk_head(3, P3{});
        step(b1, 3, a_ptr(3, buf_of(3)), std::false_type{});
        chunk_tail(P3{});
#pragma unroll 1
        for (int s = 4; s < steps; s += 4) {
            chunk_head(s, P0{});
            load_patch(a1, s + 2);
            load_patch(b1, s + 3);
            step(a0, s, a_ptr(s, buf_of(s)), std::false_type{});
            chunk_tail(P0{});
            chunk_head(s + 1, P1{});
            step(b0, s + 1, a_ptr(s + 1, buf_of(s + 1)), std::false_type{});
            chunk_tail(P1{});
            chunk_head(s + 2, P2{});
            load_patch(a0, s + 4 < last ? s + 4 : last);     // clamped at the end: re-reads the last patch, dropped
            load_patch(b0, s + 5 < last ? s + 5 : last);
            step(a1, s + 2, a_ptr(s + 2, buf_of(s + 2)), std::false_type{});
            chunk_tail(P2{});
            chunk_head(s + 3, P3{});
            step(b1, s + 3, a_ptr(s + 3, buf_of(s + 3)), std::false_type{});
            chunk_tail(P3{});
        }

        // epilogue.  Share of Y = A^T M A from the rows of M this half owns: with c_a = (M[a][0] + M[a][1] + M[a][2],
        // M[a][1] - M[a][2] - M[a][3]):  half 0: Y0 = c_0 + c_1, Y1 = c_1;  half 1: Y0 = c_2, Y1 = -c_2 - c_3.
        float part[MT][4][4];        // [tile][r][2x2]
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float ca[2][2];
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {
                    ca[a2][0] = acc[4 * a2 + 0][m][r] + acc[4 * a2 + 1][m][r] + acc[4 * a2 + 2][m][r];
                    ca[a2][1] = acc[4 * a2 + 1][m][r] - acc[4 * a2 + 2][m][r] - acc[4 * a2 + 3][m][r];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    part[m][r][0 * 2 + j] = HALF == 0 ? ca[0][j] + ca[1][j] : ca[0][j];
                    part[m][r][1 * 2 + j] = HALF == 0 ? ca[1][j] : -ca[0][j] - ca[1][j];
                }
            }
        // swap: this half finalises r = 2 HALF, 2 HALF + 1 and sends the other two
        float *outbox = c.x_s + (size_t)(threadIdx.x >> 6) * (8 * MT * 64);
        const float *inbox = c.x_s + (size_t)((threadIdx.x >> 6) ^ 4) * (8 * MT * 64);
        __syncthreads();             // the partner has read what the previous tile group left here
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int f = 0; f < 4; ++f) outbox[((m * 2 + rr) * 4 + f) * 64 + c.lane] = part[m][2 * (1 - HALF) + rr][f];
        __syncthreads();
        const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * HALF + rr;
            float yy[MT][2][2];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int f = 0; f < 4; ++f) yy[m][f >> 1][f & 1] = part[m][r][f] + inbox[((m * 2 + rr) * 4 + f) * 64 + c.lane];
            if (EPI == 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int ch = c.slice * (16 * MT) + m * 16 + 4 * g + r;
                    if (!(valid && ch < Cout)) continue;
                    float *o = c.y + (((size_t)n * Cout + ch) * H + 2 * th) * W + 2 * tw;
                    const bool h1 = 2 * th + 1 < H;
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
            } else {
                // max-feature-map pairs: (first half, second half) of pair slot q; q = MT / 2 is the mixed tile of an odd MT
                constexpr int HT = MT / 2, NQ = HT + (MT & 1);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float va[2][2], vb[2][2];
                    bool mine = true;
                    int p;
                    if (q < HT) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) { va[i][j] = yy[q][i][j]; vb[i][j] = yy[q + HT][i][j]; }
                        p = q * 16 + 4 * g + r;
                    } else {
                        // rows j < 8 (lanes 0-31) carry first halves, rows j >= 8 (lanes 32-63) the second halves of the same 8 pairs
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float own = yy[MT - 1][i][j];
                                va[i][j] = own;
                                vb[i][j] = __shfl_xor(own, 32);
                            }
                        mine = g < 2;
                        p = HT * 16 + ((4 * g + r) & 7);
                    }
                    const int ch = c.slice * split_pairs_per_slice(MT) + p;
                    const bool live = mine && ch < Cout;
                    const int chs = live ? ch : 0;
                    const float ba = c.bias ? c.bias[chs] : 0.0f, bb = c.bias ? c.bias[chs + Cout] : 0.0f;
                    if (EPI == 1) {
                        int code;
                        float vbest = pool_select(va[0][0] + ba, vb[0][0] + bb, va[0][1] + ba, vb[0][1] + bb,
                                                  va[1][0] + ba, vb[1][0] + bb, va[1][1] + ba, vb[1][1] + bb, code);
                        if (c.bn_mean) vbest = (vbest - c.bn_mean[chs]) * c.bn_invstd[chs];
                        if (valid && live && th < Ho && tw < Wo) {
                            const size_t o = ((size_t)n * Cout + ch) * Ho * Wo + (size_t)th * Wo + tw;
                            c.y[o] = vbest;
                            c.idx[o] = (uint8_t)code;
                        }
                    } else {
                        const float mu = c.bn_mean ? c.bn_mean[chs] : 0.0f, sc = c.bn_mean ? c.bn_invstd[chs] : 1.0f;
                        float out[2][2];
                        uint32_t bits = 0;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float xa = va[i][j] + ba, xb = vb[i][j] + bb;
                                const bool tb = mfm_takes_b(xa, xb);
                                bits |= (uint32_t)tb << (2 * i + j);
                                out[i][j] = ((tb ? xb : xa) - mu) * sc;
                            }
                        if (valid && live) {
                            float *o = c.y + (((size_t)n * Cout + ch) * H + 2 * th) * W + 2 * tw;
                            const bool h1 = 2 * th + 1 < H, w1 = 2 * tw + 1 < W;
                            o[0] = out[0][0];
                            if (w1) o[1] = out[0][1];
                            if (h1) {
                                o[W] = out[1][0];
                                if (w1) o[W + 1] = out[1][1];
                            }
                            c.idx[((size_t)n * Cout + ch) * TH * TW + (size_t)th * TW + tw] = (uint8_t)bits;
                        }
                    }
                }
            }
        }
    }
}

// grid = slices * ranges workgroups of 8 waves = 4 pairs; the workgroups that share a tile range (one per slice) get block
// indices that are multiples of 8 apart, i.e. the same XCD and its L2.
template <int EPI, bool STREAM, int SRC, int MT>
__global__ __launch_bounds__(kSplitThreads) void wino3x3_split_kernel(
    const float *__restrict__ x, const uint8_t *__restrict__ xsel, const float *__restrict__ U, const float *__restrict__ bias,
    const float *__restrict__ bn_mean, const float *__restrict__ bn_invstd, float *__restrict__ y, uint8_t *__restrict__ idx,
    int N, int K, int H, int W, int Cout, int slices, int ranges) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    SplitCtx<EPI, STREAM, SRC, MT> c;
    {
        const int b = blockIdx.x, per = 8 * slices, full = (ranges / 8) * per;
        if (b < full) { c.slice = (b % per) / 8; c.range = (b / per) * 8 + (b & 7); }
        else { const int r = b - full; c.slice = r % slices; c.range = (ranges / 8) * 8 + r / slices; }
    }
    c.chunks = K / split_chunk_cin(MT);
    c.steps = K / 4;
    c.Usl = U + (int64_t)c.slice * c.chunks * split_chunk_floats(MT);
    c.x_s = lds;
    c.u_s = lds + split_exchange_floats(MT);
    c.bias = bias, c.bn_mean = bn_mean, c.bn_invstd = bn_invstd, c.y = y, c.idx = idx;
    c.N = N, c.K = K, c.H = H, c.W = W, c.Cout = Cout, c.ranges = ranges;
    const int wave = threadIdx.x >> 6;
    c.pair = wave & 3;
    c.lane = threadIdx.x & 63, c.g = c.lane >> 4, c.nl = c.lane & 15;
    c.TH = (H + 1) >> 1, c.TW = (W + 1) >> 1;
    c.tiles = N * c.TH * c.TW, c.groups = (c.tiles + 15) >> 4;
    c.iters = (c.groups + ranges * kSplitPairs - 1) / (ranges * kSplitPairs);       // the same for every workgroup
    c.plane = (uint32_t)(H * W);
    c.Hs = H >> 1, c.Ws = W >> 1, c.Cs = K >> 1;
    c.cplane = (uint32_t)(c.Hs * c.Ws);
    const size_t src_elems = SRC == 0 ? (size_t)N * K * c.plane : (size_t)N * c.Cs * c.cplane;
    c.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)(src_elems * 4), 0x00020000);
    c.sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(SRC == 1 ? xsel : reinterpret_cast<const uint8_t *>(x)), 0,
                                             (int)src_elems, 0x00020000);
    // the two halves run the same number of barriers; everything position-dependent is a compile-time constant per half
    if (wave < 4) split_run<EPI, STREAM, SRC, MT, 0>(c);
    else split_run<EPI, STREAM, SRC, MT, 1>(c);
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

// Which kernel runs a layer, decided from its shape alone so that prepare and launch always agree.
//   units: forward = max-feature-map pairs C (pairs = true); input-gradient = output rows Cin (pairs = false).
// ADVSTEP_WINO_SPLIT (read once per process): 0 = the 32-rows-per-wave kernel everywhere, 1 (default) = the position-split
// kernel where it covers the rows with MT >= 3 accumulator tiles, 2 = the position-split kernel everywhere.
struct WinoPlan {
    bool split;
    int MT, slices;
};
inline int wino_split_mode() {
    static const int mode = [] {
        const char *e = getenv("ADVSTEP_WINO_SPLIT");
        return e && *e ? atoi(e) : 1;
    }();
    return mode;
}
inline WinoPlan plan_wino(int64_t units, bool pairs) {
    WinoPlan narrow{false, 2, (int)ceil_div(units, pairs ? 16 : 32)};
    const int mode = wino_split_mode();
    if (mode == 0) return narrow;
    WinoPlan best = narrow;
    int64_t best_cost = INT64_MAX;
    for (int MT = 4; MT >= 2; --MT) {                      // fewest matrix-instruction rows; ties go to the larger MT
        const int per = pairs ? split_pairs_per_slice(MT) : 16 * MT;
        const int64_t sl = ceil_div(units, per);
        if (sl * MT < best_cost) {
            best_cost = sl * MT;
            best = WinoPlan{true, MT, (int)sl};
        }
    }
    if (best.MT == 2 && mode != 2) return narrow;
    return best;
}

template <int EPI, int SRC>
int launch_wino_narrow(const float *x, const uint8_t *xsel, const float *U, const float *bias, const float *bn_mean,
                       const float *bn_invstd, float *y, uint8_t *idx, int64_t N, int64_t K, int64_t H, int64_t W, int64_t Cout,
                       int slices, hipStream_t st) {
    const int cus = 256;
    const int chunks = (int)(K / kChunkCin);
    const bool stream = chunks > kMaxResident;
    int ranges = cus / slices;
    const int64_t groups = ceil_div(N * ((H + 1) / 2) * ((W + 1) / 2), 16);
    if ((int64_t)ranges * kWaves > groups) ranges = (int)ceil_div(groups, kWaves);
    if (ranges < 1) ranges = 1;
    const size_t lds = (size_t)(stream ? 2 : chunks) * kChunkFloats * sizeof(float);
    const dim3 grid((unsigned)(slices * ranges)), block(kThreads);
    auto go = [&](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, grid, block, lds, st, x, xsel, U, bias, bn_mean, bn_invstd, y, idx, (int)N, (int)K, (int)H,
                           (int)W, (int)Cout, slices, ranges);
    };
    if (stream) go(wino3x3_kernel<EPI, true, SRC>);
    else go(wino3x3_kernel<EPI, false, SRC>);
    return status_after_launch();
}

template <int EPI, int SRC, int MT>
int launch_wino_split(const float *x, const uint8_t *xsel, const float *U, const float *bias, const float *bn_mean,
                      const float *bn_invstd, float *y, uint8_t *idx, int64_t N, int64_t K, int64_t H, int64_t W, int64_t Cout,
                      int slices, hipStream_t st) {
    const int cus = 256;
    const int chunks = (int)(K / split_chunk_cin(MT));
    const bool stream = chunks > split_max_resident(MT);
    int ranges = cus / slices;
    const int64_t groups = ceil_div(N * ((H + 1) / 2) * ((W + 1) / 2), 16);
    if ((int64_t)ranges * kSplitPairs > groups) ranges = (int)ceil_div(groups, kSplitPairs);
    if (ranges < 1) ranges = 1;
    const size_t lds = ((size_t)(stream ? 2 : chunks) * split_chunk_floats(MT) + split_exchange_floats(MT)) * sizeof(float);
    const dim3 grid((unsigned)(slices * ranges)), block(kSplitThreads);
    auto go = [&](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, grid, block, lds, st, x, xsel, U, bias, bn_mean, bn_invstd, y, idx, (int)N, (int)K, (int)H,
                           (int)W, (int)Cout, slices, ranges);
    };
    if (stream) go(wino3x3_split_kernel<EPI, true, SRC, MT>);
    else go(wino3x3_split_kernel<EPI, false, SRC, MT>);
    return status_after_launch();
}

// units / pairs as plan_wino
template <int EPI, int SRC>
int launch_wino(const float *x, const uint8_t *xsel, const float *U, const float *bias, const float *bn_mean,
                const float *bn_invstd, float *y, uint8_t *idx, int64_t N, int64_t K, int64_t H, int64_t W, int64_t Cout,
                int64_t units, hipStream_t st) {
    const WinoPlan plan = plan_wino(units, EPI != 0);
    if (!plan.split)
        return launch_wino_narrow<EPI, SRC>(x, xsel, U, bias, bn_mean, bn_invstd, y, idx, N, K, H, W, Cout, plan.slices, st);
    switch (plan.MT) {
        case 4: return launch_wino_split<EPI, SRC, 4>(x, xsel, U, bias, bn_mean, bn_invstd, y, idx, N, K, H, W, Cout, plan.slices, st);
        case 3: return launch_wino_split<EPI, SRC, 3>(x, xsel, U, bias, bn_mean, bn_invstd, y, idx, N, K, H, W, Cout, plan.slices, st);
        default: return launch_wino_split<EPI, SRC, 2>(x, xsel, U, bias, bn_mean, bn_invstd, y, idx, N, K, H, W, Cout, plan.slices, st);
    }
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
    if (!advstep_conv3x3_supported(Cin, Cout) || (mode != 0 && mode != 1)) return 0;
    const int64_t K = mode == 0 ? Cin : Cout;
    const WinoPlan plan = plan_wino(mode == 0 ? Cout / 2 : Cin, mode == 0);
    if (plan.split) return (size_t)(plan.slices * (K / split_chunk_cin(plan.MT)) * split_chunk_floats(plan.MT));
    return (size_t)(plan.slices * (K / kChunkCin) * kChunkFloats);
}

int advstep_conv3x3_prepare_f32(const float *weight, const float *gscale, float *U, int64_t Cin, int64_t Cout, int mode,
                                advstep_stream_t stream) {
    WINO_REQUIRE(weight && U && advstep_conv3x3_supported(Cin, Cout) && (mode == 0 || mode == 1));
    WINO_REQUIRE(gscale == nullptr || mode == 1);
    const int K = (int)(mode == 0 ? Cin : Cout);
    const WinoPlan plan = plan_wino(mode == 0 ? Cout / 2 : Cin, mode == 0);
    if (plan.split) {
        const int chunks = K / split_chunk_cin(plan.MT);
        const int total = plan.slices * 16 * plan.MT * K;      // every (row, k) slot of the buffer is written (zeros where dead)
        hipLaunchKernelGGL(wino_prepare_split_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                           weight, gscale, U, (int)Cin, (int)Cout, mode, plan.slices, chunks, plan.MT);
        return status_after_launch();
    }
    const int total = plan.slices * 32 * K;
    hipLaunchKernelGGL(wino_prepare_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), weight,
                       gscale, U, (int)Cin, (int)Cout, mode, plan.slices, K / kChunkCin);
    return status_after_launch();
}

int advstep_conv3x3_mfm_pool2_forward_f32(const float *x, const float *U, const float *bias, const float *bn_mean,
                                          const float *bn_invstd, float *y, uint8_t *idx, int64_t N, int64_t Cin, int64_t C,
                                          int64_t H, int64_t W, advstep_stream_t stream) {
    WINO_REQUIRE(N >= 0 && H >= 0 && W >= 0 && advstep_conv3x3_supported(Cin, 2 * C));
    if (N == 0 || H / 2 == 0 || W / 2 == 0) return ADVSTEP_OK;
    WINO_REQUIRE(x && U && y && idx && (bn_mean == nullptr) == (bn_invstd == nullptr));
    WINO_REQUIRE((uint64_t)N * Cin * H * W * 4 < (1ull << 31) && (uint64_t)N * ((H + 1) / 2) * ((W + 1) / 2) < (1ull << 31));
    return launch_wino<1, 0>(x, nullptr, U, bias, bn_mean, bn_invstd, y, idx, N, Cin, H, W, C, C, as_stream(stream));
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
    return launch_wino<2, 0>(x, nullptr, U, bias, bn_mean, bn_invstd, y, sel, N, Cin, H, W, C, C, as_stream(stream));
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
    return launch_wino<0, 0>(gout, nullptr, U, nullptr, nullptr, nullptr, gx, nullptr, N, Cout, H, W, Cin, Cin, as_stream(stream));
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
    return launch_wino<0, 1>(gy, idx, U, nullptr, nullptr, nullptr, gx, nullptr, N, 2 * C, H, W, Cin, Cin, as_stream(stream));
}

}  // extern "C"
