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

template <int CIN>
__global__ __launch_bounds__(kBlock) void conv1x1_mfm_forward_kernel(const float *__restrict__ x,
                                                                     const float *__restrict__ weight,
                                                                     const float *__restrict__ bias,
                                                                     float *__restrict__ y,
                                                                     unsigned long long *__restrict__ sel, int C,
                                                                     int64_t P, int64_t PW) {
    const int64_t n = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = p < P;
    const float *xn = x + n * CIN * P + (valid ? p : 0);
    float xr[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) xr[ci] = valid ? xn[(int64_t)ci * P] : 0.0f;

    float *yn = y + n * (int64_t)C * P + p;
    unsigned long long *sn = sel + n * (int64_t)C * PW + (p >> 6);
    for (int c = 0; c < C; ++c) {
        const float *wa = weight + (int64_t)c * CIN;  // wave-uniform: scalar loads
        const float *wb = weight + (int64_t)(c + C) * CIN;
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            a = fmaf(wa[ci], xr[ci], a);
            b = fmaf(wb[ci], xr[ci], b);
        }
        if (bias) {
            a += bias[c];
            b += bias[c + C];
        }
        const bool tb = mfm_takes_b(a, b);
        const unsigned long long word = __ballot(valid && tb);
        if (valid) {
            yn[(int64_t)c * P] = tb ? b : a;
            if ((threadIdx.x & 63) == 0) sn[(int64_t)c * PW] = word;
        }
    }
}

template <int CIN>
__global__ __launch_bounds__(kBlock) void conv1x1_mfm_backward_kernel(const float *__restrict__ gy,
                                                                      const unsigned long long *__restrict__ sel,
                                                                      const float *__restrict__ weight,
                                                                      float *__restrict__ gx, int C, int64_t P,
                                                                      int64_t PW) {
    const int64_t n = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const float *gn = gy + n * (int64_t)C * P + p;
    const unsigned long long *sn = sel + n * (int64_t)C * PW + (p >> 6);
    const int lane = threadIdx.x & 63;
    float acc[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float g = gn[(int64_t)c * P];
        const bool tb = (sn[(int64_t)c * PW] >> lane) & 1ull;
        const float ga = tb ? 0.0f : g, gb = tb ? g : 0.0f;
        const float *wa = weight + (int64_t)c * CIN;
        const float *wb = weight + (int64_t)(c + C) * CIN;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            acc[ci] = fmaf(ga, wa[ci], acc[ci]);
            acc[ci] = fmaf(gb, wb[ci], acc[ci]);
        }
    }
    float *xn = gx + n * CIN * P + p;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) xn[(int64_t)ci * P] = acc[ci];
}

constexpr int64_t kMaxGridY = 65535;

template <int CIN>
void launch_fwd(const float *x, const float *w, const float *b, float *y, unsigned long long *sel, int64_t N, int64_t C,
                int64_t P, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(P, kBlock), (unsigned)N);
    hipLaunchKernelGGL(conv1x1_mfm_forward_kernel<CIN>, grid, dim3(kBlock), 0, st, x, w, b, y, sel, (int)C, P,
                       ceil_div(P, 64));
}
template <int CIN>
void launch_bwd(const float *gy, const unsigned long long *sel, const float *w, float *gx, int64_t N, int64_t C,
                int64_t P, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(P, kBlock), (unsigned)N);
    hipLaunchKernelGGL(conv1x1_mfm_backward_kernel<CIN>, grid, dim3(kBlock), 0, st, gy, sel, w, gx, (int)C, P,
                       ceil_div(P, 64));
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

int advstep_conv1x1_mfm_forward_f32(const float *x, const float *weight, const float *bias, float *y, uint64_t *sel,
                                    int64_t N, int64_t Cin, int64_t C, int64_t P, advstep_stream_t stream) {
    C11_REQUIRE(N >= 0 && C >= 0 && P >= 0 && advstep_conv1x1_mfm_supported(Cin));
    if (N == 0 || C == 0 || P == 0) return ADVSTEP_OK;
    C11_REQUIRE(x && weight && y && sel && N <= kMaxGridY && C <= INT32_MAX && ((reinterpret_cast<uintptr_t>(sel) & 7u) == 0));
    auto *s64 = reinterpret_cast<unsigned long long *>(sel);
    hipStream_t st = as_stream(stream);
    switch (Cin) {
        case 32: launch_fwd<32>(x, weight, bias, y, s64, N, C, P, st); break;
        case 48: launch_fwd<48>(x, weight, bias, y, s64, N, C, P, st); break;
        default: launch_fwd<64>(x, weight, bias, y, s64, N, C, P, st); break;
    }
    return status_after_launch();
}

int advstep_conv1x1_mfm_backward_f32(const float *gy, const uint64_t *sel, const float *weight, float *gx, int64_t N,
                                     int64_t Cin, int64_t C, int64_t P, advstep_stream_t stream) {
    C11_REQUIRE(N >= 0 && C >= 0 && P >= 0 && advstep_conv1x1_mfm_supported(Cin));
    if (N == 0 || P == 0) return ADVSTEP_OK;
    C11_REQUIRE(gx && N <= kMaxGridY && C <= INT32_MAX);
    hipStream_t st = as_stream(stream);
    if (C == 0)
        return hipMemsetAsync(gx, 0, (size_t)N * Cin * P * sizeof(float), st) == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH;
    C11_REQUIRE(gy && sel && weight);
    auto *s64 = reinterpret_cast<const unsigned long long *>(sel);
    switch (Cin) {
        case 32: launch_bwd<32>(gy, s64, weight, gx, N, C, P, st); break;
        case 48: launch_bwd<48>(gy, s64, weight, gx, N, C, P, st); break;
        default: launch_bwd<64>(gy, s64, weight, gx, N, C, P, st); break;
    }
    return status_after_launch();
}

}  // extern "C"
