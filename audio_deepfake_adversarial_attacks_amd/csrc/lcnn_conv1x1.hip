// lcnn_conv1x1.hip — LCNN's 1x1 "network-in-network" blocks fused on gfx950:
//     Conv2d(Cin, 2C, (1, 1)) -> MaxFeatureMap2D          (src/models/lcnn.py:125-126, 132-133, 139-140, 146-147)
// forward and input-backward (C ABI: include/advstep_lcnn.h).
//
// A 1x1 convolution over NCHW is a skinny GEMM per pixel column (K = Cin = 32..64, M = 2C = 64..128) whose output
// is consumed only by the max-feature-map.  Run separately (MIOpen GEMM + ATen add + max) the 2C-channel tensor is
// written and read twice (L3 at B = 128: 265 MB each way); fused, a thread keeps its pixel's Cin inputs in
// registers, forms both halves of each channel pair with wave-uniform (scalar) weight operands, and writes only the
// C-channel winner plus ONE selection bit (wave ballot).  Reads Cin*4 B and writes C*4 B per pixel: HBM-bound
// (VALU work 2*Cin*C fma per pixel stays under the memory time at these sizes); fp32 MFMA would run at the same
// rate as the VALU here and needs no help from it.
//   forward : thread = pixel, x[:, p] in Cin registers, loop over the C pairs (2*Cin fma with SGPR weights).
//   backward: thread = pixel, Cin accumulators; for every pair the gradient goes to the selected half's weight row.
// Fixed summation order (ci ascending / c ascending), fma: deterministic; equals MIOpen's result to float rounding.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_lcnn.h"

namespace {

constexpr int kBlock = 256;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

__device__ __forceinline__ bool mfm_takes_b(float a, float b) { return !(a != a) && !(a >= b); }

// grid (ceil(P / 256), N, Z): blockIdx.z owns the channel pairs [z * cper, (z + 1) * cper) — small feature maps
// (P = 500 in LCNN's last blocks) would otherwise run one wave per SIMD with nothing to hide the scalar weight loads.
// bn_mean / bn_invstd (C, nullable): the eval-mode BatchNorm2d(affine=False) that follows every 1x1 block in LCNN,
// y = (max - mean[c]) * invstd[c], applied in the epilogue.
// PIX pixels per thread (p, p + 256, ...): every scalar weight operand is used PIX times, which halves the scalar-load
// traffic and latency per fma for the large feature maps.
template <int CIN, int PIX>
__global__ __launch_bounds__(kBlock) void conv1x1_mfm_forward_kernel(const float *__restrict__ x,
                                                                     const float *__restrict__ weight,
                                                                     const float *__restrict__ bias,
                                                                     const float *__restrict__ bn_mean,
                                                                     const float *__restrict__ bn_invstd,
                                                                     float *__restrict__ y,
                                                                     unsigned long long *__restrict__ sel, int C, int cper,
                                                                     int64_t P, int64_t PW) {
    const int64_t n = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * (kBlock * PIX) + threadIdx.x;
    bool valid[PIX];
    float xr[PIX][CIN];
#pragma unroll
    for (int q = 0; q < PIX; ++q) {
        const int64_t p = p0 + (int64_t)q * kBlock;
        valid[q] = p < P;
        const float *xn = x + n * CIN * P + (valid[q] ? p : 0);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) xr[q][ci] = valid[q] ? xn[(int64_t)ci * P] : 0.0f;
    }
    float *yn = y + n * (int64_t)C * P + p0;
    unsigned long long *sn = sel + n * (int64_t)C * PW + (p0 >> 6);
    const int c_begin = blockIdx.z * cper;
    const int c_end = c_begin + cper < C ? c_begin + cper : C;
    for (int c = c_begin; c < c_end; ++c) {
        const float *wa = weight + (int64_t)c * CIN;  // wave-uniform: scalar loads
        const float *wb = weight + (int64_t)(c + C) * CIN;
        float a[PIX], b[PIX];
#pragma unroll
        for (int q = 0; q < PIX; ++q) a[q] = b[q] = 0.0f;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float ua = wa[ci], ub = wb[ci];
#pragma unroll
            for (int q = 0; q < PIX; ++q) {
                a[q] = fmaf(ua, xr[q][ci], a[q]);
                b[q] = fmaf(ub, xr[q][ci], b[q]);
            }
        }
        const float ba = bias ? bias[c] : 0.0f, bb = bias ? bias[c + C] : 0.0f;
        const float mu = bn_mean ? bn_mean[c] : 0.0f, is = bn_mean ? bn_invstd[c] : 1.0f;
#pragma unroll
        for (int q = 0; q < PIX; ++q) {
            const float va = bias ? a[q] + ba : a[q], vb = bias ? b[q] + bb : b[q];
            const bool tb = mfm_takes_b(va, vb);
            const unsigned long long word = __ballot(valid[q] && tb);
            float v = tb ? vb : va;
            if (bn_mean) v = (v - mu) * is;
            if (valid[q]) {
                yn[(int64_t)c * P + (int64_t)q * kBlock] = v;
                if ((threadIdx.x & 63) == 0) sn[(int64_t)c * PW + q * (kBlock / 64)] = word;
            }
        }
    }
}

// grid (ceil(P / (256 PIX)), N, CIN / CHUNK): blockIdx.z owns input channels [z * CHUNK, (z + 1) * CHUNK).
// gscale (C, nullable): the following BatchNorm's backward, gy * invstd[c].
template <int CIN, int CHUNK, int PIX>
__global__ __launch_bounds__(kBlock) void conv1x1_mfm_backward_kernel(const float *__restrict__ gy,
                                                                      const unsigned long long *__restrict__ sel,
                                                                      const float *__restrict__ weight,
                                                                      const float *__restrict__ gscale,
                                                                      float *__restrict__ gx, int C, int64_t P,
                                                                      int64_t PW) {
    const int64_t n = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * (kBlock * PIX) + threadIdx.x;
    if (p0 >= P) return;
    const int ci0 = blockIdx.z * CHUNK;
    const float *gn = gy + n * (int64_t)C * P + p0;
    const unsigned long long *sn = sel + n * (int64_t)C * PW + (p0 >> 6);
    const int lane = threadIdx.x & 63;
    bool valid[PIX];
#pragma unroll
    for (int q = 0; q < PIX; ++q) valid[q] = p0 + (int64_t)q * kBlock < P;
    float acc[PIX][CHUNK];
#pragma unroll
    for (int q = 0; q < PIX; ++q)
#pragma unroll
        for (int k = 0; k < CHUNK; ++k) acc[q][k] = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float sc = gscale ? gscale[c] : 1.0f;
        float ga[PIX], gb[PIX];
#pragma unroll
        for (int q = 0; q < PIX; ++q) {
            float g = valid[q] ? gn[(int64_t)c * P + (int64_t)q * kBlock] : 0.0f;
            if (gscale) g *= sc;
            const bool tb = valid[q] && ((sn[(int64_t)c * PW + q * (kBlock / 64)] >> lane) & 1ull);
            ga[q] = tb ? 0.0f : g;
            gb[q] = tb ? g : 0.0f;
        }
        const float *wa = weight + (int64_t)c * CIN + ci0;
        const float *wb = weight + (int64_t)(c + C) * CIN + ci0;
#pragma unroll
        for (int k = 0; k < CHUNK; ++k) {
            const float ua = wa[k], ub = wb[k];
#pragma unroll
            for (int q = 0; q < PIX; ++q) {
                acc[q][k] = fmaf(ga[q], ua, acc[q][k]);
                acc[q][k] = fmaf(gb[q], ub, acc[q][k]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PIX; ++q) {
        if (!valid[q]) continue;
        float *xn = gx + n * CIN * P + (int64_t)ci0 * P + p0 + (int64_t)q * kBlock;
#pragma unroll
        for (int k = 0; k < CHUNK; ++k) xn[(int64_t)k * P] = acc[q][k];
    }
}

constexpr int64_t kMaxGridY = 65535;
constexpr int64_t kWantBlocks = 2048;  // ~8 workgroups per CU keep the scalar-load latency covered

template <int CIN, int PIX>
void launch_fwd_pix(const float *x, const float *w, const float *b, const float *bn_mean, const float *bn_invstd, float *y,
                    unsigned long long *sel, int64_t N, int64_t C, int64_t P, int64_t z, hipStream_t st) {
    const int64_t cper = ceil_div(C, z);
    z = ceil_div(C, cper);
    const dim3 grid((unsigned)ceil_div(P, kBlock * PIX), (unsigned)N, (unsigned)z);
    hipLaunchKernelGGL((conv1x1_mfm_forward_kernel<CIN, PIX>), grid, dim3(kBlock), 0, st, x, w, b, bn_mean, bn_invstd, y,
                       sel, (int)C, (int)cper, P, ceil_div(P, 64));
}
template <int CIN, int BIGPIX>
void launch_fwd(const float *x, const float *w, const float *b, const float *bn_mean, const float *bn_invstd, float *y,
                unsigned long long *sel, int64_t N, int64_t C, int64_t P, hipStream_t st) {
    const int64_t blocks = ceil_div(P, kBlock) * N;
    if (blocks >= 2 * kWantBlocks) {  // plenty of parallelism: amortise the scalar weight loads over BIGPIX pixels
        launch_fwd_pix<CIN, BIGPIX>(x, w, b, bn_mean, bn_invstd, y, sel, N, C, P, 1, st);
        return;
    }
    int64_t z = blocks >= kWantBlocks ? 1 : ceil_div(kWantBlocks, blocks);
    if (z > 8) z = 8;
    if (z > C) z = C;
    launch_fwd_pix<CIN, 1>(x, w, b, bn_mean, bn_invstd, y, sel, N, C, P, z, st);
}
template <int CIN, int CHUNK, int PIX>
void launch_bwd_chunk(const float *gy, const unsigned long long *sel, const float *w, const float *gscale, float *gx,
                      int64_t N, int64_t C, int64_t P, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(P, kBlock * PIX), (unsigned)N, (unsigned)(CIN / CHUNK));
    hipLaunchKernelGGL((conv1x1_mfm_backward_kernel<CIN, CHUNK, PIX>), grid, dim3(kBlock), 0, st, gy, sel, w, gscale, gx,
                       (int)C, P, ceil_div(P, 64));
}
template <int CIN, int SMALL, int BIGPIX>
void launch_bwd(const float *gy, const unsigned long long *sel, const float *w, const float *gscale, float *gx, int64_t N,
                int64_t C, int64_t P, hipStream_t st) {
    const int64_t blocks = ceil_div(P, kBlock) * N;
    if (blocks >= 2 * kWantBlocks)
        launch_bwd_chunk<CIN, CIN, BIGPIX>(gy, sel, w, gscale, gx, N, C, P, st);
    else if (blocks >= kWantBlocks)
        launch_bwd_chunk<CIN, CIN, 1>(gy, sel, w, gscale, gx, N, C, P, st);
    else
        launch_bwd_chunk<CIN, SMALL, 1>(gy, sel, w, gscale, gx, N, C, P, st);
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
    return (size_t)N * (size_t)C * (size_t)ceil_div(P, 64) * sizeof(unsigned long long);
}

int advstep_conv1x1_mfm_forward_f32(const float *x, const float *weight, const float *bias, const float *bn_mean,
                                    const float *bn_invstd, float *y, uint64_t *sel, int64_t N, int64_t Cin, int64_t C,
                                    int64_t P, advstep_stream_t stream) {
    C11_REQUIRE(N >= 0 && C >= 0 && P >= 0 && advstep_conv1x1_mfm_supported(Cin));
    if (N == 0 || C == 0 || P == 0) return ADVSTEP_OK;
    C11_REQUIRE(x && weight && y && sel && N <= kMaxGridY && C <= INT32_MAX && ((reinterpret_cast<uintptr_t>(sel) & 7u) == 0));
    C11_REQUIRE((bn_mean == nullptr) == (bn_invstd == nullptr));
    auto *s64 = reinterpret_cast<unsigned long long *>(sel);
    hipStream_t st = as_stream(stream);
    switch (Cin) {
        case 32: launch_fwd<32, 2>(x, weight, bias, bn_mean, bn_invstd, y, s64, N, C, P, st); break;
        case 48: launch_fwd<48, 2>(x, weight, bias, bn_mean, bn_invstd, y, s64, N, C, P, st); break;
        default: launch_fwd<64, 1>(x, weight, bias, bn_mean, bn_invstd, y, s64, N, C, P, st); break;
    }
    return status_after_launch();
}

int advstep_conv1x1_mfm_backward_f32(const float *gy, const uint64_t *sel, const float *weight, const float *gscale,
                                     float *gx, int64_t N, int64_t Cin, int64_t C, int64_t P, advstep_stream_t stream) {
    C11_REQUIRE(N >= 0 && C >= 0 && P >= 0 && advstep_conv1x1_mfm_supported(Cin));
    if (N == 0 || P == 0) return ADVSTEP_OK;
    C11_REQUIRE(gx && N <= kMaxGridY && C <= INT32_MAX);
    hipStream_t st = as_stream(stream);
    if (C == 0)
        return hipMemsetAsync(gx, 0, (size_t)N * Cin * P * sizeof(float), st) == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH;
    C11_REQUIRE(gy && sel && weight);
    auto *s64 = reinterpret_cast<const unsigned long long *>(sel);
    switch (Cin) {
        case 32: launch_bwd<32, 8, 2>(gy, s64, weight, gscale, gx, N, C, P, st); break;
        case 48: launch_bwd<48, 12, 2>(gy, s64, weight, gscale, gx, N, C, P, st); break;
        default: launch_bwd<64, 16, 1>(gy, s64, weight, gscale, gx, N, C, P, st); break;
    }
    return status_after_launch();
}

}  // extern "C"
