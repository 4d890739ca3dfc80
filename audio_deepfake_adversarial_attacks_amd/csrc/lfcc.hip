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
// The DCT projection as a true dense contraction on the matrix cores (fp32-in / fp32-accumulate MFMA, exact f32):
//   out (frames x 80) = max(band_db, floor) (frames x 128) . dct (128 x 80)           K = 128
//   g   (frames x 128) = dout (frames x 80) . dct^T (80 x 128)                         K = 80
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the 32x32
// accumulator lives in 16 registers, col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// One wave owns 32 frames; a 256-thread workgroup owns 128.  Operands are staged in LDS with odd row pitches so the
// per-step fragment reads are bank-conflict free.  Specialised for M = 128, K = 80 (LFCC); other sizes use the VALU
// kernels above.
// ---------------------------------------------------------------------------------------------------------

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMfmaFrames = 128;   // frames per workgroup (4 waves x 32)
constexpr int kLfccM = 128, kLfccK = 80, kLfccKPad = 96;

__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(kBlock) void lfcc_project_mfma_kernel(const float *__restrict__ band_db,
                                                                   const float *__restrict__ dct, float *stats,
                                                                   float top_db, float *__restrict__ out, int NF) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float(*band_s)[kLfccM + 1] = reinterpret_cast<float(*)[kLfccM + 1]>(smem);                      // [128][129]
    float(*dct_s)[kLfccKPad] = reinterpret_cast<float(*)[kLfccKPad]>(smem + kMfmaFrames * (kLfccM + 1));  // [128][96]
    const int64_t b = blockIdx.y;
    const int t0 = blockIdx.x * kMfmaFrames;
    const int nfr = NF - t0 < kMfmaFrames ? NF - t0 : kMfmaFrames;
    const float gmax = stats[0];
    const float floor_db = gmax - top_db;
    const float *src = band_db + (b * NF + t0) * kLfccM;
    int ties = 0;
    // stage the (frames x 128) band tile and the DCT with 16-byte global loads, several in flight per thread
    // all 16 loads of a thread are issued before the first is used (a load -> wait -> LDS write loop is one HBM latency
    // per iteration: 16 of them were the whole kernel time); rows beyond the tile read a clamped address and are zeroed
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    constexpr int kStage = kMfmaFrames * kLfccM / 4 / kBlock;   // 16
    const int lim = nfr * kLfccM / 4;
    float4 st[kStage];
#pragma unroll
    for (int j = 0; j < kStage; ++j) {
        const int i = threadIdx.x + j * kBlock;
        st[j] = src4[i < lim ? i : lim - 1];
    }
#pragma unroll
    for (int j = 0; j < kStage; ++j) {
        const int i = threadIdx.x + j * kBlock;
        float4 v = st[j];
        if (i < lim) {
            ties += (v.x == gmax) + (v.y == gmax) + (v.z == gmax) + (v.w == gmax);
            float *pv = &v.x;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                pv[e] = (pv[e] != pv[e]) ? pv[e] : ((floor_db != floor_db) ? floor_db : (pv[e] > floor_db ? pv[e] : floor_db));
        } else {
            v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        float *row = &band_s[i >> 5][(i & 31) * 4];
        row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
    }
    const float4 *dct4 = reinterpret_cast<const float4 *>(dct);
    constexpr int kDctStage = kLfccM * kLfccKPad / 4 / kBlock;   // 12
    float4 dt[kDctStage];
#pragma unroll
    for (int j = 0; j < kDctStage; ++j) {
        const int i = threadIdx.x + j * kBlock, m = i / (kLfccKPad / 4), k4 = i - m * (kLfccKPad / 4);
        dt[j] = dct4[m * (kLfccK / 4) + (k4 < kLfccK / 4 ? k4 : 0)];
    }
#pragma unroll
    for (int j = 0; j < kDctStage; ++j) {
        const int i = threadIdx.x + j * kBlock, m = i / (kLfccKPad / 4), k4 = i - m * (kLfccKPad / 4);
        *reinterpret_cast<float4 *>(&dct_s[m][k4 * 4]) = k4 < kLfccK / 4 ? dt[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (ties) atomicAdd(&stats[1], (float)ties);
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = wave * 32;
    if (row0 < nfr) {  // wave-uniform
        f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0};
        const int li = lane & 31, lk = lane >> 5;
#pragma unroll 4
        for (int s = 0; s < kLfccM / 2; ++s) {
            const float a = band_s[row0 + li][2 * s + lk];
            const float *br = dct_s[2 * s + lk];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, br[li], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, br[32 + li], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, br[64 + li], acc2, 0, 0, 0);
        }
        float *dst = out + (b * NF + t0 + row0) * kLfccK;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma_row(r, lane);
            if (row0 + row < nfr) {
                float *o = dst + (int64_t)row * kLfccK;
                o[li] = acc0[r];
                o[32 + li] = acc1[r];
                if (li < kLfccK - 64) o[64 + li] = acc2[r];
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void lfcc_project_backward_mfma_kernel(const float *__restrict__ dout,
                                                                            const float *__restrict__ dct,
                                                                            const float *__restrict__ band_db,
                                                                            float *stats, float top_db,
                                                                            float *__restrict__ dband, int NF) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float(*g_s)[kLfccK + 1] = reinterpret_cast<float(*)[kLfccK + 1]>(smem);                           // [128][81]
    float(*dct_s)[kLfccK + 1] = reinterpret_cast<float(*)[kLfccK + 1]>(smem + kMfmaFrames * (kLfccK + 1));  // [128][81]
    float(*band_s)[kLfccM + 1] =
        reinterpret_cast<float(*)[kLfccM + 1]>(smem + (kMfmaFrames + kLfccM) * (kLfccK + 1));               // [128][129]
    __shared__ float red[kBlock / 64];
    const int64_t b = blockIdx.y;
    const int t0 = blockIdx.x * kMfmaFrames;
    const int nfr = NF - t0 < kMfmaFrames ? NF - t0 : kMfmaFrames;
    const float *src = dout + (b * NF + t0) * kLfccK;
    // staging in batches: every load of a batch is issued before the first is used (see the forward kernel)
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    const float4 *dct4 = reinterpret_cast<const float4 *>(dct);
    const float4 *band4 = reinterpret_cast<const float4 *>(band_db + (b * NF + t0) * kLfccM);
    constexpr int kGStage = kMfmaFrames * kLfccK / 4 / kBlock;   // 10 (dout tile and the DCT have the same size)
    constexpr int kBStage = kMfmaFrames * kLfccM / 4 / kBlock;   // 16
    static_assert(kMfmaFrames == kLfccM, "the dout tile and the DCT are staged with the same index map");
    {
        const int lim = nfr * kLfccK / 4;
        float4 gq[kGStage], dq[kGStage];
#pragma unroll
        for (int j = 0; j < kGStage; ++j) {
            const int i = threadIdx.x + j * kBlock;
            gq[j] = src4[i < lim ? i : lim - 1];
            dq[j] = dct4[i];
        }
#pragma unroll
        for (int j = 0; j < kGStage; ++j) {
            const int i = threadIdx.x + j * kBlock, t = i / (kLfccK / 4), k4 = i - t * (kLfccK / 4);
            const float4 v = i < lim ? gq[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float *row = &g_s[t][k4 * 4];
            row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
            float *drow = &dct_s[t][k4 * 4];
            drow[0] = dq[j].x; drow[1] = dq[j].y; drow[2] = dq[j].z; drow[3] = dq[j].w;
        }
    }
    {
        const int lim = nfr * kLfccM / 4;
        float4 bq[kBStage];
#pragma unroll
        for (int j = 0; j < kBStage; ++j) {
            const int i = threadIdx.x + j * kBlock;
            bq[j] = band4[i < lim ? i : lim - 1];
        }
#pragma unroll
        for (int j = 0; j < kBStage; ++j) {
            const int i = threadIdx.x + j * kBlock;
            if (i < lim) {
                float *row = &band_s[i >> 5][(i & 31) * 4];
                row[0] = bq[j].x; row[1] = bq[j].y; row[2] = bq[j].z; row[3] = bq[j].w;
            }
        }
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = wave * 32;
    const float floor_db = stats[0] - top_db;
    float floored = 0.0f;
    if (row0 < nfr) {
        f32x16 acc[4] = {{0}, {0}, {0}, {0}};
        const int li = lane & 31, lk = lane >> 5;
#pragma unroll 4
        for (int s = 0; s < kLfccK / 2; ++s) {
            const int k = 2 * s + lk;
            const float a = g_s[row0 + li][k];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dct_s[32 * j + li][k], acc[j], 0, 0, 0);
        }
        float *dst = dband + (b * NF + t0 + row0) * kLfccM;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma_row(r, lane);
            if (row0 + row < nfr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = 32 * j + li;
                    const float db = band_s[row0 + row][m];
                    const float sgrad = acc[j][r];
                    const float wgt = db > floor_db ? 1.0f : (db == floor_db ? 0.5f : 0.0f);
                    floored += (1.0f - wgt) * sgrad;
                    dst[(int64_t)row * kLfccM + m] = (wgt * sgrad) * dlog_of_db(db);
                }
            }
        }
    }
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
    if (K == kLfccK && M == kLfccM) {  // matrix-core path
        const dim3 mgrid((unsigned)ceil_div(NF, kMfmaFrames), (unsigned)B);
        const size_t lds = (size_t)(kMfmaFrames * (kLfccM + 1) + kLfccM * kLfccKPad) * sizeof(float);
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(lfcc_project_mfma_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)attr;  // > 64 KiB of dynamic LDS must be opted into once
        hipLaunchKernelGGL(lfcc_project_mfma_kernel, mgrid, dim3(kBlock), lds, st, band_db, dct, stats, top_db, out, (int)NF);
    } else if (K == 80)
        hipLaunchKernelGGL(lfcc_project_kernel<20>, grid, dim3(kBlock), 0, st, band_db, dct, stats, top_db, out, (int)M, (int)NF);
    else if (K == 40)
        hipLaunchKernelGGL(lfcc_project_kernel<10>, grid, dim3(kBlock), 0, st, band_db, dct, stats, top_db, out, (int)M, (int)NF);
    else
        hipLaunchKernelGGL(lfcc_project_kernel<5>, grid, dim3(kBlock), 0, st, band_db, dct, stats, top_db, out, (int)M, (int)NF);
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
    if (K == kLfccK && M == kLfccM) {  // matrix-core path
        const dim3 mgrid((unsigned)ceil_div(NF, kMfmaFrames), (unsigned)B);
        const size_t lds =
            (size_t)(kMfmaFrames * (kLfccK + 1) + kLfccM * (kLfccK + 1) + kMfmaFrames * (kLfccM + 1)) * sizeof(float);
        static const hipError_t attr = hipFuncSetAttribute(
            reinterpret_cast<const void *>(lfcc_project_backward_mfma_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)attr;
        hipLaunchKernelGGL(lfcc_project_backward_mfma_kernel, mgrid, dim3(kBlock), lds, st, dout, dct, band_db, stats, top_db, dband, (int)NF);
    } else if (K == 80)
        hipLaunchKernelGGL(lfcc_project_backward_kernel<20>, grid, dim3(kBlock), 0, st, dout, dct, band_db, stats, top_db, dband, (int)M, (int)NF);
    else if (K == 40)
        hipLaunchKernelGGL(lfcc_project_backward_kernel<10>, grid, dim3(kBlock), 0, st, dout, dct, band_db, stats, top_db, dband, (int)M, (int)NF);
    else
        hipLaunchKernelGGL(lfcc_project_backward_kernel<5>, grid, dim3(kBlock), 0, st, dout, dct, band_db, stats, top_db, dband, (int)M, (int)NF);
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
