// lfcc.hip — the LFCC frontend after the STFT, fused for gfx950 (C ABI: include/advstep_frontend.h).
// See the header for the op chain it replaces.  All kernels stream (B, 128..257, 404) f32 planes with the frame
// index fastest (coalesced), the DCT / its transpose run on the VALU with wave-uniform (scalar) coefficient operands.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_frontend.h"

namespace {

constexpr int kBlock = 256;
constexpr int kFrames = 64;   // frames per projection tile (one wave per coefficient chunk)
constexpr float kAmin = 1e-10f;
constexpr float kDbScale = 4.342944819032518f;  // 10 / ln(10)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

__device__ __forceinline__ float max_nan(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    return a > b ? a : b;
}

__device__ __forceinline__ float block_max(float v, float *lds) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max_nan(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = lds[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = max_nan(r, lds[w]);
    return r;
}

__device__ __forceinline__ float block_sum(float v, float *lds) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = lds[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += lds[w];
    return r;
}

// d/d band of 10*log10(clamp(band, amin)), band recovered from its dB value
__device__ __forceinline__ float dlog_of_db(float db) {
    return (db > -100.0f) ? kDbScale / expf(db * 0.23025850929940457f) : 0.0f;
}

// Layouts (the STFT's native one: torch.stft returns a (B, F, NF) VIEW of a (B, NF, F) buffer): everything here is
// frame-major with the spectral index fastest — spec (B, NF, F) complex, band_db / dband (B, NF, M), out / dout (B, NF, K).

// thread = (frame t, band m), m fastest: neighbouring lanes read neighbouring bins
__global__ __launch_bounds__(kBlock) void lfcc_bands_kernel(const float2 *__restrict__ spec,
                                                            const int32_t *__restrict__ fb_start,
                                                            const float *__restrict__ fb_w, int span,
                                                            float *__restrict__ band_db, float *__restrict__ bmax, int F,
                                                            int M, int NF) {
    __shared__ float lds[kBlock / 64];
    const int64_t b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float db = -INFINITY;
    if (i < M * NF) {
        const int t = i / M, m = i - t * M;
        const int f0 = fb_start[m];
        const float2 *row = spec + (b * NF + t) * F;
        float band = 0.0f;
        for (int j = 0; j < span; ++j) {
            const int f = f0 + j;
            if (f < F) {
                const float2 z = row[f];
                band = fmaf(fb_w[m * span + j], fmaf(z.x, z.x, z.y * z.y), band);
            }
        }
        db = 10.0f * log10f(band > kAmin ? band : (band != band ? band : kAmin));
        band_db[(b * NF) * M + i] = db;
    }
    const float r = block_max(db, lds);
    if (threadIdx.x == 0) bmax[b * gridDim.x + blockIdx.x] = r;
}

// one workgroup of 1024 threads, 16-byte loads: ~26 K block maxima at B = 128
__global__ __launch_bounds__(1024) void lfcc_reduce_max_kernel(const float *__restrict__ bmax, int64_t n,
                                                               float *__restrict__ stats) {
    __shared__ float lds[1024 / 64];
    float v = -INFINITY;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(bmax) & 15u) == 0) ? n / 4 : 0;
    const float4 *b4 = reinterpret_cast<const float4 *>(bmax);
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
        const float4 q = b4[i];
        v = max_nan(max_nan(v, q.x), max_nan(max_nan(q.y, q.z), q.w));
    }
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 1024) v = max_nan(v, bmax[i]);
    v = block_max(v, lds);
    if (threadIdx.x == 0) {
        stats[0] = v;
        stats[1] = 0.0f;
        stats[2] = 0.0f;
        stats[3] = 0.0f;
    }
}

constexpr int kMaxBands = 128;

// grid (ceil(NF / 64), B), block 256 = 64 frames x 4 coefficient chunks (one wave per chunk: uniform DCT operands).
// The 64 x M band tile is contiguous in memory: staged through LDS (floor applied on the way in), rows padded by one
// word so the per-frame reads of a wave hit 32 different banks.
template <int KC>  // coefficients per chunk (K = 4 * KC)
__global__ __launch_bounds__(kBlock) void lfcc_project_kernel(const float *__restrict__ band_db,
                                                              const float *__restrict__ dct, float *stats, float top_db,
                                                              float *__restrict__ out, int M, int NF) {
    constexpr int K = 4 * KC;
    __shared__ float band_s[kFrames][kMaxBands + 1];
    __shared__ float tile[kFrames][K + 1];
    const int64_t b = blockIdx.y;
    const int t0 = blockIdx.x * kFrames;
    const int nfr = NF - t0 < kFrames ? NF - t0 : kFrames;
    const int tl = threadIdx.x & 63, kc = threadIdx.x >> 6;
    const float gmax = stats[0];
    const float floor_db = gmax - top_db;
    const float *src = band_db + (b * NF + t0) * M;
    int ties = 0;
    for (int i = threadIdx.x; i < nfr * M; i += kBlock) {
        float v = src[i];
        ties += (v == gmax) ? 1 : 0;
        v = (v != v) ? v : ((floor_db != floor_db) ? floor_db : (v > floor_db ? v : floor_db));  // torch.max, NaN kept
        band_s[i / M][i % M] = v;
    }
    if (ties) atomicAdd(&stats[1], (float)ties);
    __syncthreads();
    float acc[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) acc[k] = 0.0f;
    if (tl < nfr) {
        for (int m = 0; m < M; ++m) {
            const float v = band_s[tl][m];
            const float *dr = dct + m * K + kc * KC;
#pragma unroll
            for (int k = 0; k < KC; ++k) acc[k] = fmaf(v, dr[k], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) tile[tl][kc * KC + k] = acc[k];
    __syncthreads();
    float *dst = out + (b * NF + t0) * K;
    for (int i = threadIdx.x; i < nfr * K; i += kBlock) dst[i] = tile[i / K][i % K];
}

// grid (ceil(NF / 64), B), block 256 = 64 frames x 4 band chunks; band tile staged in LDS and overwritten in place
template <int KC>
__global__ __launch_bounds__(kBlock) void lfcc_project_backward_kernel(const float *__restrict__ dout,
                                                                       const float *__restrict__ dct,
                                                                       const float *__restrict__ band_db, float *stats,
                                                                       float top_db, float *__restrict__ dband, int M,
                                                                       int NF) {
    constexpr int K = 4 * KC;
    __shared__ float tile[kFrames][K + 1];
    __shared__ float band_s[kFrames][kMaxBands + 1];
    __shared__ float lds[kBlock / 64];
    const int64_t b = blockIdx.y;
    const int t0 = blockIdx.x * kFrames;
    const int nfr = NF - t0 < kFrames ? NF - t0 : kFrames;
    const int tl = threadIdx.x & 63, mc = threadIdx.x >> 6;
    const float *src = dout + (b * NF + t0) * K;
    for (int i = threadIdx.x; i < nfr * K; i += kBlock) tile[i / K][i % K] = src[i];
    const float *bsrc = band_db + (b * NF + t0) * M;
    for (int i = threadIdx.x; i < nfr * M; i += kBlock) band_s[i / M][i % M] = bsrc[i];
    __syncthreads();
    float g[K];
#pragma unroll
    for (int k = 0; k < K; ++k) g[k] = tile[tl][k];
    const float floor_db = stats[0] - top_db;
    const int mper = (M + 3) / 4;
    float floored = 0.0f;
    if (tl < nfr) {
        for (int m = mc * mper; m < (mc + 1) * mper && m < M; ++m) {
            const float *dr = dct + m * K;
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) s = fmaf(g[k], dr[k], s);
            const float db = band_s[tl][m];
            const float wgt = db > floor_db ? 1.0f : (db == floor_db ? 0.5f : 0.0f);  // torch.max(a, b) backward
            floored += (1.0f - wgt) * s;
            band_s[tl][m] = (wgt * s) * dlog_of_db(db);
        }
    }
    floored = block_sum(floored, lds);  // (contains the barrier that orders the in-place tile writes)
    if (threadIdx.x == 0 && floored != 0.0f) atomicAdd(&stats[2], floored);
    float *dst = dband + (b * NF + t0) * M;
    for (int i = threadIdx.x; i < nfr * M; i += kBlock) dst[i] = band_s[i / M][i % M];
}

// ---------------------------------------------------------------------------------------------------------
// The DCT projection as a dense contraction on the matrix cores (fp32-in / fp32-accumulate MFMA, exact f32 products):
//   out (frames x 80) = max(band_db, floor) (frames x 128) . dct (128 x 80)           K = 128
//   g   (frames x 128) = dout (frames x 80) . dct^T (80 x 128)                         K = 80
// v_mfma_f32_16x16x4_f32: lane l = (g = l >> 4, j = l & 15) supplies A[j][k = g] and B[k = g][n = j]; the 16 x 16 result sits
// in 4 registers, D[4 g + r][j].  (B, NF) is one flat frame axis (the floor is batch-wide, nothing is per utterance).
//
// Shape of both kernels (round 4; measured steps in profiles/r04_lfcc_project_experiments.txt):
//  * 64 frames and FOUR waves per workgroup, <= 34 KB of LDS and <= 128 registers: four workgroups per CU, so the 808
//    workgroups of a B = 128 batch are resident at once, one wave of each on every SIMD.  (Five waves per workgroup - one per
//    16 output columns - measured 14 us for 8.5 us of matrix instructions: every workgroup's fifth wave lands on the SIMD of
//    its first.  Rounds 1 - 3 staged both operands of a 128-frame tile in 115 / 150 KB: one workgroup per CU, loads, matrix
//    instructions and stores one after the other - 29 / 46 us.)
//  * The DCT side of a wave is constant over its contraction and lives in REGISTERS, loaded as 16-byte pieces of a fragment
//    table (advstep_lfcc_project_prepare_f32: the DCT re-ordered once per weight version so that a lane's values for four
//    consecutive k-steps are one float4 and a wave's request is 1 KB contiguous).  Reading the (128, 80) matrix itself in
//    operand order is one lane per cycle in the texture addresser - 16 rows x 16 bytes per dword load, 64 cycles each: 10 us
//    of a 36 us backward kernel went there.
//  * The frame side streams from an LDS tile all four waves share - staged with 16-byte loads / stores at a pitch of
//    (row length + 4) words, which makes the fragment read tile[16 rb + j][4 s + g] hit 64 different banks; one base register,
//    immediate offsets.
//  * Results leave through the same LDS tile: accumulators -> tile (conflict-free at pitch 84 / 132) -> whole rows as 16-byte
//    stores (64-byte pieces straight from the accumulator layout cost 4.7 / 8.5 us of stores).
// Forward: wave w owns output columns 16 w .. 16 w + 15 for all four 16-frame row blocks, and columns 64 .. 79 for row block w
// (5 accumulators, 160 matrix instructions per wave, 2 x 32 DCT registers).  Backward: wave w owns bands 32 w .. 32 w + 31
// (8 accumulators, 160 matrix instructions, 2 x 20 DCT registers).
// Specialised for M = 128, K = 80 (LFCC); other sizes use the VALU kernels above.
// ---------------------------------------------------------------------------------------------------------

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLfccM = 128, kLfccK = 80;
constexpr int kPTile = 64, kPThreads = 256;                          // frames / threads per workgroup
constexpr int kBandPitch = kLfccM + 4, kCepPitch = kLfccK + 4;       // 132, 84 words
constexpr int kFragFloats = kLfccM * kLfccK;                         // per direction

// frag[0 .. MK): forward,  F[c][q][lane][e] = dct[4 (4 q + e) + g][16 c + j],  c < 5, q < 8
// frag[MK .. 2 MK): backward, G[c][q][lane][e] = dct[16 c + j][4 (4 q + e) + g],  c < 8, q < 5
__global__ __launch_bounds__(kBlock) void lfcc_project_prepare_kernel(const float *__restrict__ dct, float *__restrict__ frag) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= 2 * kFragFloats) return;
    const bool bwd = i >= kFragFloats;
    const int r = bwd ? i - kFragFloats : i;
    const int e = r & 3, lane = (r >> 2) & 63, j = lane & 15, g = lane >> 4, cq = r >> 8;
    if (!bwd) {
        const int c = cq / 8, q = cq % 8;
        frag[i] = dct[(4 * (4 * q + e) + g) * kLfccK + 16 * c + j];
    } else {
        const int c = cq / 5, q = cq % 5;
        frag[i] = dct[(16 * c + j) * kLfccK + 4 * (4 * q + e) + g];
    }
}

// grid ceil(F / 64), F = B NF frames; block 256.  The batch maximum is reduced here, by every workgroup, from the `nblk`
// per-workgroup maxima the band kernel left (13 KB at B = 128: L2-resident, the reads ride under the tile's own loads) - no
// one-workgroup reduction launch between the two kernels; workgroup 0 publishes stats = {max, 0, 0, 1}: the tie count is left to
// the backward pass (stats[3] == 1 tells it so), which sees every dB value anyway.
__global__ __launch_bounds__(kPThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void lfcc_project_mfma_kernel(const float *__restrict__ band_db,
                                                                     const float *__restrict__ frag,
                                                                     const float *block_maxima, int nblk,
                                                                     float *stats, float top_db, float *__restrict__ out,
                                                                     int64_t F) {
    __shared__ __attribute__((aligned(16))) float tile[kPTile * kBandPitch];     // 33 792 B: the dB tile, then the cepstra
    __shared__ float red[kPThreads / 64];
    const int64_t f0 = (int64_t)blockIdx.x * kPTile;
    const int nfr = F - f0 < kPTile ? (int)(F - f0) : kPTile;
    // the block maxima first (loads return in order: their reduction then runs while the tile is still in flight), in batches
    // of 8 independent loads - a plain strided loop is one L2 round trip per iteration
    float v = -INFINITY;
    for (int base = 0; base < nblk; base += 8 * kPThreads) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kPThreads + threadIdx.x;
            t[u] = block_maxima[i < nblk ? i : nblk - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v = max_nan(v, t[u]);
    }
    // the (frames x 128) tile: every load of a thread is in flight before the first is used; rows beyond the last frame read
    // a clamped address and are zeroed
    const float4 *src4 = reinterpret_cast<const float4 *>(band_db + f0 * kLfccM);
    constexpr int kStage = kPTile * kLfccM / 4 / kPThreads;   // 8
    const int lim = nfr * (kLfccM / 4);
    float4 st[kStage];
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
        const int i = threadIdx.x + u * kPThreads;
        st[u] = src4[i < lim ? i : lim - 1];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lj = lane & 15, lg = lane >> 4;
    // DCT fragments: columns 16 w .. (own) and 64 .. 79 (shared out by row block)
    float4 bo[kLfccM / 16], bs[kLfccM / 16];
    {
        const float4 *f4 = reinterpret_cast<const float4 *>(frag);
#pragma unroll
        for (int q = 0; q < kLfccM / 16; ++q) {
            bo[q] = f4[(wave * (kLfccM / 16) + q) * 64 + lane];
            bs[q] = f4[(4 * (kLfccM / 16) + q) * 64 + lane];
        }
    }
    const float gmax = block_max(v, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // (large batches hand in stats itself as a one-entry array of maxima - not restrict-qualified for that reason: the
        // other workgroups are reading stats[0] then, so it is not stored again)
        if (block_maxima != stats) stats[0] = gmax;
        stats[1] = 0.0f;
        stats[2] = 0.0f;
        stats[3] = 1.0f;
    }
    const float floor_db = gmax - top_db;
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
        const int i = threadIdx.x + u * kPThreads;
        float4 q = st[u];
        if (i < lim) {
            float *pv = &q.x;
#pragma unroll
            for (int e = 0; e < 4; ++e)      // torch.max(x, floor): NaN kept
                pv[e] = (pv[e] != pv[e]) ? pv[e] : ((floor_db != floor_db) ? floor_db : (pv[e] > floor_db ? pv[e] : floor_db));
        } else {
            q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        *reinterpret_cast<float4 *>(&tile[(i >> 5) * kBandPitch + (i & 31) * 4]) = q;
    }
    __syncthreads();

    f32x4 acc[4], accs = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float *ap = tile + lj * kBandPitch + lg;
    const float *aw = ap + wave * 16 * kBandPitch;             // this wave's row block, for the shared columns
#pragma unroll
    for (int q = 0; q < kLfccM / 16; ++q) {
        const float bov[4] = {bo[q].x, bo[q].y, bo[q].z, bo[q].w}, bsv[4] = {bs[q].x, bs[q].y, bs[q].z, bs[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = 4 * q + e;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[rb * 16 * kBandPitch + 4 * s], bov[e], acc[rb], 0, 0, 0);
            accs = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[4 * s], bsv[e], accs, 0, 0, 0);
        }
    }
    __syncthreads();                                           // every wave is done reading the dB tile
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) tile[(16 * rb + 4 * lg + r) * kCepPitch + 16 * wave + lj] = acc[rb][r];
        tile[(16 * wave + 4 * lg + r) * kCepPitch + 64 + lj] = accs[r];
    }
    __syncthreads();
    float4 *dst4 = reinterpret_cast<float4 *>(out + f0 * kLfccK);
    constexpr int kOut4 = kPTile * kLfccK / 4;                 // 1280 = 5 x 256
    const int olim = nfr * (kLfccK / 4);
#pragma unroll
    for (int u = 0; u < kOut4 / kPThreads; ++u) {
        const int i = threadIdx.x + u * kPThreads;
        const int t = i / (kLfccK / 4), k4 = i - t * (kLfccK / 4);
        if (i < olim) dst4[i] = *reinterpret_cast<const float4 *>(&tile[t * kCepPitch + 4 * k4]);
    }
}

// grid ceil(F / 64); block 256.  `zero` (may be null): zero_n floats this launch also zero-fills - the waveform gradient the
// overlap-add kernel behind it accumulates into (33 MB of stores that ride under this kernel's loads instead of a memset node of
// their own).  stats[3] == 1 (left by the forward kernel above): count the elements equal to the batch maximum into stats[1].
__global__ __launch_bounds__(kPThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void lfcc_project_backward_mfma_kernel(const float *__restrict__ dout,
                                                                              const float *__restrict__ frag,
                                                                              const float *__restrict__ band_db,
                                                                              float *stats, float top_db,
                                                                              float *__restrict__ dband, int64_t F,
                                                                              float *__restrict__ zero, int64_t zero_n) {
    __shared__ __attribute__((aligned(16))) float tile[kPTile * kBandPitch];     // the dout tile [64][84], then dband [64][132]
    __shared__ float red[kPThreads / 64];
    const int64_t f0 = (int64_t)blockIdx.x * kPTile;
    const int nfr = F - f0 < kPTile ? (int)(F - f0) : kPTile;
    const float4 *src4 = reinterpret_cast<const float4 *>(dout + f0 * kLfccK);
    constexpr int kIn4 = kPTile * kLfccK / 4;                  // 1280 = 5 x 256
    const int lim = nfr * (kLfccK / 4);
    float4 st[kIn4 / kPThreads];
#pragma unroll
    for (int u = 0; u < kIn4 / kPThreads; ++u) {
        const int i = threadIdx.x + u * kPThreads;
        st[u] = src4[i < lim ? i : lim - 1];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lj = lane & 15, lg = lane >> 4;
    float4 bq[2][kLfccK / 16];
    {
        const float4 *f4 = reinterpret_cast<const float4 *>(frag + kFragFloats);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int q = 0; q < kLfccK / 16; ++q) bq[cc][q] = f4[((2 * wave + cc) * (kLfccK / 16) + q) * 64 + lane];
    }
    if (zero) {
        const int64_t gid = (int64_t)blockIdx.x * kPThreads + threadIdx.x, total = (int64_t)gridDim.x * kPThreads;
        if ((reinterpret_cast<uintptr_t>(zero) & 15u) == 0) {
            float4 *z4 = reinterpret_cast<float4 *>(zero);
            const int64_t n4 = zero_n / 4;
            for (int64_t i = gid; i < n4; i += total) z4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int64_t i = n4 * 4 + gid; i < zero_n; i += total) zero[i] = 0.0f;
        } else {
            for (int64_t i = gid; i < zero_n; i += total) zero[i] = 0.0f;
        }
    }
#pragma unroll
    for (int u = 0; u < kIn4 / kPThreads; ++u) {
        const int i = threadIdx.x + u * kPThreads;
        const int t = i / (kLfccK / 4), k4 = i - t * (kLfccK / 4);
        *reinterpret_cast<float4 *>(&tile[t * kCepPitch + k4 * 4]) = i < lim ? st[u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // the dB rows the epilogue needs, whole rows as 16-byte loads: requested before the products
    const float4 *band4 = reinterpret_cast<const float4 *>(band_db + f0 * kLfccM);
    constexpr int kOut4 = kPTile * kLfccM / 4 / kPThreads;     // 8
    const int blim = nfr * (kLfccM / 4);
    float4 db[kOut4];
#pragma unroll
    for (int u = 0; u < kOut4; ++u) {
        const int i = threadIdx.x + u * kPThreads;
        db[u] = band4[i < blim ? i : blim - 1];
    }
    __syncthreads();

    f32x4 acc[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb][0] = acc[rb][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float *ap = tile + lj * kCepPitch + lg;
#pragma unroll
    for (int q = 0; q < kLfccK / 16; ++q) {
        const float b0[4] = {bq[0][q].x, bq[0][q].y, bq[0][q].z, bq[0][q].w};
        const float b1[4] = {bq[1][q].x, bq[1][q].y, bq[1][q].z, bq[1][q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = 4 * q + e;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const float a = ap[rb * 16 * kCepPitch + 4 * s];
                acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0[e], acc[rb][0], 0, 0, 0);
                acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1[e], acc[rb][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                           // every wave is done reading the dout tile
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                tile[(16 * rb + 4 * lg + r) * kBandPitch + 16 * (2 * wave + cc) + lj] = acc[rb][cc][r];
    __syncthreads();
    const float gmax = stats[0], floor_db = gmax - top_db;
    const bool count_ties = stats[3] == 1.0f;
    float floored = 0.0f;
    int ties = 0;
    float4 *dst4 = reinterpret_cast<float4 *>(dband + f0 * kLfccM);
#pragma unroll
    for (int u = 0; u < kOut4; ++u) {
        const int i = threadIdx.x + u * kPThreads;
        if (i < blim) {
            const float4 sg = *reinterpret_cast<const float4 *>(&tile[(i >> 5) * kBandPitch + (i & 31) * 4]);
            const float sv[4] = {sg.x, sg.y, sg.z, sg.w}, dv[4] = {db[u].x, db[u].y, db[u].z, db[u].w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = dv[e];
                ties += (d == gmax) ? 1 : 0;
                const float wgt = d > floor_db ? 1.0f : (d == floor_db ? 0.5f : 0.0f);   // torch.max(a, b) backward
                floored += (1.0f - wgt) * sv[e];
                o[e] = (wgt * sv[e]) * dlog_of_db(d);
            }
            dst4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (count_ties && ties) atomicAdd(&stats[1], (float)ties);
    floored = block_sum(floored, red);
    if (threadIdx.x == 0 && floored != 0.0f) atomicAdd(&stats[2], floored);
}

__global__ __launch_bounds__(kBlock) void lfcc_floor_fixup_kernel(const float *__restrict__ band_db,
                                                                  const float *__restrict__ stats,
                                                                  float *__restrict__ dband, int64_t n) {
    const float s = stats[2];
    if (s == 0.0f) return;  // nothing was floored: the usual case
    const float gmax = stats[0];
    const float share = s / (stats[1] > 0.0f ? stats[1] : 1.0f);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float db = band_db[i];
        if (db == gmax) dband[i] += share * dlog_of_db(db);
    }
}

// thread = (frame t, bin f), f fastest
__global__ __launch_bounds__(kBlock) void lfcc_bands_backward_kernel(const float *__restrict__ dband,
                                                                     const float2 *__restrict__ spec,
                                                                     const int32_t *__restrict__ fbt_start,
                                                                     const float *__restrict__ fbt_w, int span_t,
                                                                     float2 *__restrict__ dspec, int F, int M, int NF,
                                                                     int hermitian_half) {
    const int64_t b = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= F * NF) return;
    const int t = i / F, f = i - t * F;
    const int m0 = fbt_start[f];
    const float *row = dband + (b * NF + t) * M;
    float dp = 0.0f;
    for (int j = 0; j < span_t; ++j) {
        const int m = m0 + j;
        if (m < M) dp = fmaf(fbt_w[f * span_t + j], row[m], dp);
    }
    const float2 z = spec[(b * NF) * F + i];
    float2 g = make_float2(2.0f * z.x * dp, 2.0f * z.y * dp);
    if (hermitian_half) {
        // hand the gradient of a one-sided real FFT to a c2r transform: interior bins are counted twice there
        if (f == 0 || f == F - 1) g.y = 0.0f; else { g.x *= 0.5f; g.y *= 0.5f; }
    }
    dspec[(b * NF) * F + i] = g;
}

// frames[b, f, n] = w[n] * x[b, reflect(f * hop + n - pad)]   (torch.stft(center=True, pad_mode="reflect") framing)
__global__ __launch_bounds__(kBlock) void stft_frames_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             float *__restrict__ frames, int T, int NF, int hop,
                                                             int nfft, int pad) {
    const int64_t b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;   // over NF * nfft / 4 float4 groups
    const int per_frame = nfft >> 2;
    if (i >= (int64_t)NF * per_frame) return;
    const int f = (int)(i / per_frame), n0 = (int)(i - (int64_t)f * per_frame) * 4;
    const float *xb = x + b * T;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int q = f * hop + n0 + k - pad;
        q = q < 0 ? -q : q;
        q = q >= T ? 2 * (T - 1) - q : q;
        v[k] = w[n0 + k] * xb[q];
    }
    reinterpret_cast<float4 *>(frames + (b * NF + f) * nfft)[n0 >> 2] = make_float4(v[0], v[1], v[2], v[3]);
}

// transpose of the framing: dx[b, t] = sum over padded positions q that read sample t, over frames covering q
__global__ __launch_bounds__(kBlock) void stft_overlap_add_kernel(const float *__restrict__ dframes,
                                                                  const float *__restrict__ w, float *__restrict__ dx,
                                                                  int T, int NF, int hop, int nfft, int pad) {
    const int64_t b = blockIdx.y;
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= T) return;
    const float *db = dframes + b * (int64_t)NF * nfft;
    // padded coordinates p = q + pad of the (at most three) positions that read sample t
    int ps[3];
    int np = 0;
    ps[np++] = t + pad;
    if (t >= 1 && t <= pad) ps[np++] = pad - t;                       // left reflection
    if (t <= T - 2 && t >= T - 1 - pad) ps[np++] = 2 * (T - 1) - t + pad;  // right reflection
    float acc = 0.0f;
    for (int k = 0; k < np; ++k) {
        const int p = ps[k];
        int f_hi = p / hop;
        if (f_hi > NF - 1) f_hi = NF - 1;
        int f_lo = (p - nfft + hop) / hop;   // ceil((p - nfft + 1) / hop)
        if (f_lo < 0) f_lo = 0;
        for (int f = f_lo; f <= f_hi; ++f) {
            const int n = p - f * hop;
            if (n >= 0 && n < nfft) acc = fmaf(w[n], db[(int64_t)f * nfft + n], acc);
        }
    }
    dx[b * T + t] = acc;
}

constexpr int64_t kMaxGridY = 65535;

}  // namespace

#define LFCC_REQUIRE(cond) \
    do {                   \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

size_t advstep_lfcc_block_count(int64_t B, int64_t M, int64_t NF) {
    if (B <= 0 || M <= 0 || NF <= 0) return 0;
    return (size_t)B * (size_t)ceil_div(M * NF, kBlock);
}

int advstep_lfcc_bands_f32(const float *spec, const int32_t *fb_start, const float *fb_w, int64_t span, float *band_db,
                           float *block_max, int64_t B, int64_t F, int64_t M, int64_t NF, advstep_stream_t stream) {
    LFCC_REQUIRE(B >= 0 && F >= 0 && M >= 0 && NF >= 0 && span >= 1);
    if (B == 0 || M == 0 || NF == 0) return ADVSTEP_OK;
    LFCC_REQUIRE(spec && fb_start && fb_w && band_db && block_max && B <= kMaxGridY && M * NF <= INT32_MAX &&
                 F * NF <= INT32_MAX);
    const dim3 grid((unsigned)ceil_div(M * NF, kBlock), (unsigned)B);
    hipLaunchKernelGGL(lfcc_bands_kernel, grid, dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(spec), fb_start, fb_w, (int)span, band_db, block_max, (int)F,
                       (int)M, (int)NF);
    return status_after_launch();
}

int advstep_lfcc_reduce_max_f32(const float *block_max, int64_t n, float *stats, advstep_stream_t stream) {
    LFCC_REQUIRE(n >= 1 && block_max && stats);
    hipLaunchKernelGGL(lfcc_reduce_max_kernel, dim3(1), dim3(1024), 0, as_stream(stream), block_max, n, stats);
    return status_after_launch();
}

int advstep_lfcc_project_f32(const float *band_db, const float *dct, float *stats, float top_db, float *out, int64_t B,
                             int64_t M, int64_t NF, int64_t K, advstep_stream_t stream) {
    LFCC_REQUIRE(B >= 0 && M >= 0 && M <= kMaxBands && NF >= 0 && (K == 80 || K == 40 || K == 20));
    if (B == 0 || NF == 0) return ADVSTEP_OK;
    LFCC_REQUIRE(band_db && dct && stats && out && B <= kMaxGridY && M <= INT32_MAX && NF <= INT32_MAX);
    const dim3 grid((unsigned)ceil_div(NF, kFrames), (unsigned)B);
    hipStream_t st = as_stream(stream);
    if (K == 80)
        hipLaunchKernelGGL(lfcc_project_kernel<20>, grid, dim3(kBlock), 0, st, band_db, dct, stats, top_db, out, (int)M, (int)NF);
    else if (K == 40)
        hipLaunchKernelGGL(lfcc_project_kernel<10>, grid, dim3(kBlock), 0, st, band_db, dct, stats, top_db, out, (int)M, (int)NF);
    else
        hipLaunchKernelGGL(lfcc_project_kernel<5>, grid, dim3(kBlock), 0, st, band_db, dct, stats, top_db, out, (int)M, (int)NF);
    return status_after_launch();
}

size_t advstep_lfcc_project_fragment_floats(int64_t M, int64_t K) {
    return (M == kLfccM && K == kLfccK) ? (size_t)2 * kFragFloats : 0;
}

int advstep_lfcc_project_prepare_f32(const float *dct, int64_t M, int64_t K, float *frag, advstep_stream_t stream) {
    if (!(M == kLfccM && K == kLfccK)) return ADVSTEP_OK;          // no matrix-core path for this size: nothing to prepare
    LFCC_REQUIRE(dct && frag && (reinterpret_cast<uintptr_t>(frag) & 15u) == 0);
    hipLaunchKernelGGL(lfcc_project_prepare_kernel, dim3((unsigned)ceil_div(2 * kFragFloats, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), dct, frag);
    return status_after_launch();
}

int advstep_lfcc_max_project_f32(const float *band_db, const float *dct, const float *frag, const float *block_max, int64_t n,
                                 float *stats, float top_db, float *out, int64_t B, int64_t M, int64_t NF, int64_t K,
                                 advstep_stream_t stream) {
    LFCC_REQUIRE(n >= 1 && block_max && stats);
    if (!(K == kLfccK && M == kLfccM) || !frag || B <= 0 || NF <= 0 || n > INT32_MAX) {      // other sizes: the two launches
        const int st = advstep_lfcc_reduce_max_f32(block_max, n, stats, stream);
        return st != ADVSTEP_OK ? st : advstep_lfcc_project_f32(band_db, dct, stats, top_db, out, B, M, NF, K, stream);
    }
    LFCC_REQUIRE(band_db && out && B * NF <= INT32_MAX && (reinterpret_cast<uintptr_t>(frag) & 15u) == 0 &&
                 (reinterpret_cast<uintptr_t>(band_db) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0);
    // every workgroup reduces the block maxima itself: fine for the fused STFT kernel's 26 per utterance (13 KB at B = 128), not for
    // advstep_lfcc_bands_f32's 202 per utterance (100 KB x 808 workgroups) - those are reduced by one launch first and the
    // projection is handed stats[0] as a one-entry array (which it then leaves alone)
    if (n > 8192) {
        const int st = advstep_lfcc_reduce_max_f32(block_max, n, stats, stream);
        if (st != ADVSTEP_OK) return st;
        block_max = stats;
        n = 1;
    }
    hipLaunchKernelGGL(lfcc_project_mfma_kernel, dim3((unsigned)ceil_div(B * NF, kPTile)), dim3(kPThreads), 0, as_stream(stream),
                       band_db, frag, block_max, (int)n, stats, top_db, out, B * NF);
    return status_after_launch();
}

int advstep_lfcc_project_backward_f32(const float *dout, const float *dct, const float *band_db, float *stats,
                                      float top_db, float *dband, int64_t B, int64_t M, int64_t NF, int64_t K,
                                      advstep_stream_t stream) {
    LFCC_REQUIRE(B >= 0 && M >= 0 && M <= kMaxBands && NF >= 0 && (K == 80 || K == 40 || K == 20));
    if (B == 0 || NF == 0 || M == 0) return ADVSTEP_OK;
    LFCC_REQUIRE(dout && dct && band_db && stats && dband && B <= kMaxGridY && M <= INT32_MAX && NF <= INT32_MAX);
    const dim3 grid((unsigned)ceil_div(NF, kFrames), (unsigned)B);
    hipStream_t st = as_stream(stream);
    if (K == 80)
        hipLaunchKernelGGL(lfcc_project_backward_kernel<20>, grid, dim3(kBlock), 0, st, dout, dct, band_db, stats, top_db, dband, (int)M, (int)NF);
    else if (K == 40)
        hipLaunchKernelGGL(lfcc_project_backward_kernel<10>, grid, dim3(kBlock), 0, st, dout, dct, band_db, stats, top_db, dband, (int)M, (int)NF);
    else
        hipLaunchKernelGGL(lfcc_project_backward_kernel<5>, grid, dim3(kBlock), 0, st, dout, dct, band_db, stats, top_db, dband, (int)M, (int)NF);
    return status_after_launch();
}

int advstep_lfcc_project_backward_zero_f32(const float *dout, const float *dct, const float *frag, const float *band_db,
                                           float *stats, float top_db, float *dband, int64_t B, int64_t M, int64_t NF,
                                           int64_t K, float *zero, int64_t zero_n, advstep_stream_t stream) {
    LFCC_REQUIRE(zero_n >= 0 && (zero || zero_n == 0));
    if (!(K == kLfccK && M == kLfccM) || !frag || B <= 0 || NF <= 0) {             // other sizes: a memset node + the plain call
        if (zero_n > 0 && hipMemsetAsync(zero, 0, (size_t)zero_n * sizeof(float), as_stream(stream)) != hipSuccess)
            return ADVSTEP_ELAUNCH;
        return advstep_lfcc_project_backward_f32(dout, dct, band_db, stats, top_db, dband, B, M, NF, K, stream);
    }
    LFCC_REQUIRE(dout && band_db && stats && dband && B * NF <= INT32_MAX && (reinterpret_cast<uintptr_t>(frag) & 15u) == 0 &&
                 (reinterpret_cast<uintptr_t>(dout) & 15u) == 0 && (reinterpret_cast<uintptr_t>(band_db) & 15u) == 0 &&
                 (reinterpret_cast<uintptr_t>(dband) & 15u) == 0);
    hipLaunchKernelGGL(lfcc_project_backward_mfma_kernel, dim3((unsigned)ceil_div(B * NF, kPTile)), dim3(kPThreads), 0,
                       as_stream(stream), dout, frag, band_db, stats, top_db, dband, B * NF, zero_n > 0 ? zero : nullptr, zero_n);
    return status_after_launch();
}

int advstep_lfcc_floor_fixup_f32(const float *band_db, const float *stats, float *dband, int64_t n,
                                 advstep_stream_t stream) {
    LFCC_REQUIRE(n >= 0);
    if (n == 0) return ADVSTEP_OK;
    LFCC_REQUIRE(band_db && stats && dband);
    const int64_t blocks = ceil_div(n, kBlock);
    hipLaunchKernelGGL(lfcc_floor_fixup_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kBlock), 0,
                       as_stream(stream), band_db, stats, dband, n);
    return status_after_launch();
}

int advstep_lfcc_bands_backward_f32(const float *dband, const float *spec, const int32_t *fbt_start, const float *fbt_w,
                                    int64_t span_t, float *dspec, int64_t B, int64_t F, int64_t M, int64_t NF,
                                    int hermitian_half, advstep_stream_t stream) {
    LFCC_REQUIRE(B >= 0 && F >= 0 && M >= 0 && NF >= 0 && span_t >= 1);
    if (B == 0 || F == 0 || NF == 0) return ADVSTEP_OK;
    LFCC_REQUIRE(dband && spec && fbt_start && fbt_w && dspec && B <= kMaxGridY && F * NF <= INT32_MAX);
    const dim3 grid((unsigned)ceil_div(F * NF, kBlock), (unsigned)B);
    hipLaunchKernelGGL(lfcc_bands_backward_kernel, grid, dim3(kBlock), 0, as_stream(stream), dband,
                       reinterpret_cast<const float2 *>(spec), fbt_start, fbt_w, (int)span_t,
                       reinterpret_cast<float2 *>(dspec), (int)F, (int)M, (int)NF, hermitian_half);
    return status_after_launch();
}

int advstep_stft_frames_f32(const float *x, const float *window, float *frames, int64_t B, int64_t T, int64_t NF,
                            int64_t hop, int64_t nfft, advstep_stream_t stream) {
    LFCC_REQUIRE(B >= 0 && T >= 0 && NF >= 0 && hop >= 1 && nfft >= 4 && nfft % 4 == 0);
    if (B == 0 || NF == 0) return ADVSTEP_OK;
    const int64_t pad = nfft / 2;
    LFCC_REQUIRE(x && window && frames && B <= kMaxGridY && T > pad && (NF - 1) * hop + nfft - pad <= T + pad &&
                 T + nfft <= INT32_MAX && ((reinterpret_cast<uintptr_t>(frames) & 15u) == 0));
    const dim3 grid((unsigned)ceil_div(NF * (nfft / 4), kBlock), (unsigned)B);
    hipLaunchKernelGGL(stft_frames_kernel, grid, dim3(kBlock), 0, as_stream(stream), x, window, frames, (int)T, (int)NF,
                       (int)hop, (int)nfft, (int)pad);
    return status_after_launch();
}

int advstep_stft_overlap_add_f32(const float *dframes, const float *window, float *dx, int64_t B, int64_t T, int64_t NF,
                                 int64_t hop, int64_t nfft, advstep_stream_t stream) {
    LFCC_REQUIRE(B >= 0 && T >= 0 && NF >= 0 && hop >= 1 && nfft >= 4);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    const int64_t pad = nfft / 2;
    LFCC_REQUIRE(dframes && window && dx && B <= kMaxGridY && T > pad && T + nfft <= INT32_MAX);
    const dim3 grid((unsigned)ceil_div(T, kBlock), (unsigned)B);
    hipLaunchKernelGGL(stft_overlap_add_kernel, grid, dim3(kBlock), 0, as_stream(stream), dframes, window, dx, (int)T,
                       (int)NF, (int)hop, (int)nfft, (int)pad);
    return status_after_launch();
}

}  // extern "C"
