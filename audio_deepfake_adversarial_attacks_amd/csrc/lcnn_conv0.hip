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
//             pairs with the 2 x 25 taps + 2 biases as wave-uniform (scalar) operands: 200 v_fma per pair.
//   backward: thread = one 2x2 input patch; weights staged in LDS (per-lane tap lookup); no atomics.
// VALU-bound (13.2 GFLOP at B = 128), not HBM-bound; MFMA does not apply (K = 25, one input channel).
// Accumulation order is fixed (taps row-major, then channels), fma contraction allowed: deterministic, and within
// float rounding of any other convolution implementation (MIOpen's differs in the last bits as well).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_lcnn.h"

namespace {

constexpr int kBlock = 256;
constexpr int K = 5, KK = 25, PAD = 2;

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

    float *yn = y + n * (int64_t)C * Ho * Wo + p;
    uint8_t *in = idx + n * (int64_t)C * Ho * Wo + p;
    for (int c = 0; c < C; ++c) {
        const float *wa = weight + c * KK;        // wave-uniform addresses: scalar loads
        const float *wb = weight + (c + C) * KK;
        float a00, a01, a10, a11, b00, b01, b10, b11;
        a00 = a01 = a10 = a11 = 0.0f;
        b00 = b01 = b10 = b11 = 0.0f;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const float ua = wa[kh * K + kw], ub = wb[kh * K + kw];
                a00 = fmaf(ua, win[kh][kw], a00);
                a01 = fmaf(ua, win[kh][kw + 1], a01);
                a10 = fmaf(ua, win[kh + 1][kw], a10);
                a11 = fmaf(ua, win[kh + 1][kw + 1], a11);
                b00 = fmaf(ub, win[kh][kw], b00);
                b01 = fmaf(ub, win[kh][kw + 1], b01);
                b10 = fmaf(ub, win[kh + 1][kw], b10);
                b11 = fmaf(ub, win[kh + 1][kw + 1], b11);
            }
        }
        const float ba = bias ? bias[c] : 0.0f, bb = bias ? bias[c + C] : 0.0f;
        int code;
        const float v = pool_select(a00 + ba, b00 + bb, a01 + ba, b01 + bb, a10 + ba, b10 + bb, a11 + ba, b11 + bb, code);
        yn[(int64_t)c * Ho * Wo] = v;
        in[(int64_t)c * Ho * Wo] = (uint8_t)code;
    }
}

// thread = one pooled-cell footprint (2x2 input patch); gathers from the 3x3 neighbouring pooled cells.
__global__ __launch_bounds__(kBlock) void conv5_mfm_pool2_backward_kernel(const float *__restrict__ gy,
                                                                          const uint8_t *__restrict__ idx,
                                                                          const float *__restrict__ weight,
                                                                          float *__restrict__ gx, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // 2C * 25 taps
    for (int i = threadIdx.x; i < 2 * C * KK; i += kBlock) wl[i] = weight[i];
    __syncthreads();

    const int Ho = H >> 1, Wo = W >> 1;
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;  // patches cover a trailing odd row / column too
    const int64_t n = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= Hp * Wp) return;
    const int hp = p / Wp, wp = p - hp * Wp;
    const float *gn = gy + n * (int64_t)C * Ho * Wo;
    const uint8_t *in = idx + n * (int64_t)C * Ho * Wo;

    float acc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    for (int c = 0; c < C; ++c) {
        const float *gc = gn + (int64_t)c * Ho * Wo;
        const uint8_t *ic = in + (int64_t)c * Ho * Wo;
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
            const int ho = hp + dh;
            if (ho < 0 || ho >= Ho) continue;
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                const int wo = wp + dw;
                if (wo < 0 || wo >= Wo) continue;
                const int code = ic[ho * Wo + wo];
                const float g = gc[ho * Wo + wo];
                const float *wsel = wl + (c + ((code & 4) ? C : 0)) * KK;
                // winner's conv position relative to this patch's origin (2hp, 2wp)
                const int rh = 2 * dh + ((code >> 1) & 1), rw = 2 * dw + (code & 1);
                // d conv[h'] / d in[h] has tap kh = h - h' + PAD
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int kh = i - rh + PAD;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int kw = j - rw + PAD;
                        if (kh >= 0 && kh < K && kw >= 0 && kw < K) acc[i][j] = fmaf(g, wsel[kh * K + kw], acc[i][j]);
                    }
                }
            }
        }
    }
    float *xn = gx + n * (int64_t)H * W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int h = 2 * hp + i;
        if (h >= H) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int w = 2 * wp + j;
            if (w < W) xn[(int64_t)h * W + w] = acc[i][j];
        }
    }
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
    CONV0_REQUIRE(gx && N <= kMaxGridY && C <= 256 && H * W <= INT32_MAX);
    hipStream_t st = as_stream(stream);
    if (C == 0 || H / 2 == 0 || W / 2 == 0)
        return hipMemsetAsync(gx, 0, (size_t)N * H * W * sizeof(float), st) == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH;
    CONV0_REQUIRE(gy && idx && weight);
    const int64_t patches = ((H + 1) / 2) * ((W + 1) / 2);
    const dim3 grid((unsigned)ceil_div(patches, kBlock), (unsigned)N);
    const size_t lds = (size_t)2 * C * KK * sizeof(float);
    hipLaunchKernelGGL(conv5_mfm_pool2_backward_kernel, grid, dim3(kBlock), lds, st, gy, idx, weight, gx, (int)C, (int)H,
                       (int)W);
    return status_after_launch();
}

}  // extern "C"
