// lcnn_mfm.hip — gfx950 kernels for LCNN's max-feature-map (+ 2x2 max-pool), forward and input-backward
// (C ABI: include/advstep_lcnn.h).
//
// Replaces ATen's  view(N,2,C,H,W).max(1)  (reduce kernel writing values + int64 indices; backward = fill + scatter)
// and the MaxPool2d((2,2),(2,2)) that follows four of the nine MFMs (forward writes int64 indices too; backward is a
// gather-style kernel).  These are pure HBM streaming over the model's biggest activations.  Here:
//   * MFM forward reads both channel halves once (8 B/output) and writes the value plus ONE selection byte per 4
//     outputs (4.25 B) instead of value + int64 index (12 B);
//   * MFM backward writes both halves of the input gradient in one coalesced pass from gy + the selection bytes
//     (no zero-fill pass, no scatter);
//   * MFM + pool forward reads the 2x2x2 candidates of two neighbouring pooled outputs with four float4 loads and
//     writes 8 B of values + 2 index bytes; the intermediate MFM tensor never exists;
//   * MFM + pool backward writes the full-resolution gradient (one non-zero per 8 candidates) with four float4 stores.
// 16 B per lane everywhere, 2-4 independent accesses per thread in flight, grid (tiles, N).  Semantics (ties, NaN)
// follow the ATen kernels exactly; see the header.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_lcnn.h"

namespace {

constexpr int kBlock = 256;
constexpr int kGroupsPerThread = 4;  // MFM: float4 groups per thread

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

// at::native::MaxOps combine for two candidates at indices 0 (a) and 1 (b): true when b is selected.
__device__ __forceinline__ bool mfm_takes_b(float a, float b) { return !(a != a) && !(a >= b); }

// ---------------------------------------------------------------------------------------------------------
// MFM alone: per sample, P = C*HW outputs, G = ceil(P / 4) groups; thread = one group of 4 outputs
// ---------------------------------------------------------------------------------------------------------

// bias (2C) may be null; when given, a = x[n, c] + bias[c], b = x[n, c + C] + bias[c + C] (the conv's bias add,
// evaluate_... conv -> add_(bias) -> max, folded into this pass; one rounding, like ATen's separate add kernel).
template <bool VEC>
__global__ __launch_bounds__(kBlock) void mfm_forward_kernel(const float *__restrict__ x,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ bn_mean,
                                                             const float *__restrict__ bn_invstd, float *__restrict__ y,
                                                             uint8_t *__restrict__ sel, int64_t P, int64_t G,
                                                             int64_t HW, int64_t C) {
    const int64_t n = blockIdx.y;
    const float *xa = x + n * 2 * P;
    const float *xb = xa + P;
    float *yo = y + n * P;
    uint8_t *so = sel + n * G;
    const int64_t g0 = (int64_t)blockIdx.x * (kBlock * kGroupsPerThread) + threadIdx.x;
    float4 a[kGroupsPerThread], b[kGroupsPerThread];
#pragma unroll
    for (int k = 0; k < kGroupsPerThread; ++k) {
        const int64_t g = g0 + (int64_t)k * kBlock;
        if (g < G) {
            if (VEC) {
                a[k] = reinterpret_cast<const float4 *>(xa)[g];
                b[k] = reinterpret_cast<const float4 *>(xb)[g];
            } else {
                const int64_t i = g * 4;
                a[k].x = xa[i];
                b[k].x = xb[i];
                a[k].y = (i + 1 < P) ? xa[i + 1] : 0.0f;
                b[k].y = (i + 1 < P) ? xb[i + 1] : 0.0f;
                a[k].z = (i + 2 < P) ? xa[i + 2] : 0.0f;
                b[k].z = (i + 2 < P) ? xb[i + 2] : 0.0f;
                a[k].w = (i + 3 < P) ? xa[i + 3] : 0.0f;
                b[k].w = (i + 3 < P) ? xb[i + 3] : 0.0f;
            }
        }
    }
    if (bias) {
#pragma unroll
        for (int k = 0; k < kGroupsPerThread; ++k) {
            const int64_t g = g0 + (int64_t)k * kBlock;
            if (g < G) {
                const int64_t i = g * 4;
                const int64_t c0 = i / HW, c3 = (i + 3 < P ? i + 3 : P - 1) / HW;
                if (c0 == c3) {  // the usual case: the 4 outputs share a channel
                    const float ba = bias[c0], bb = bias[c0 + C];
                    a[k].x += ba; a[k].y += ba; a[k].z += ba; a[k].w += ba;
                    b[k].x += bb; b[k].y += bb; b[k].z += bb; b[k].w += bb;
                } else {
                    const int64_t c1 = (i + 1 < P ? i + 1 : P - 1) / HW, c2 = (i + 2 < P ? i + 2 : P - 1) / HW;
                    a[k].x += bias[c0]; b[k].x += bias[c0 + C];
                    a[k].y += bias[c1]; b[k].y += bias[c1 + C];
                    a[k].z += bias[c2]; b[k].z += bias[c2 + C];
                    a[k].w += bias[c3]; b[k].w += bias[c3 + C];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kGroupsPerThread; ++k) {
        const int64_t g = g0 + (int64_t)k * kBlock;
        if (g < G) {
            const bool t0 = mfm_takes_b(a[k].x, b[k].x), t1 = mfm_takes_b(a[k].y, b[k].y);
            const bool t2 = mfm_takes_b(a[k].z, b[k].z), t3 = mfm_takes_b(a[k].w, b[k].w);
            float4 o;
            o.x = t0 ? b[k].x : a[k].x;
            o.y = t1 ? b[k].y : a[k].y;
            o.z = t2 ? b[k].z : a[k].z;
            o.w = t3 ? b[k].w : a[k].w;
            if (bn_mean) {  // eval-mode BatchNorm2d(affine=False) that follows: (v - mean[c]) * invstd[c]
                const int64_t i = g * 4;
                const int64_t c0 = i / HW, c1 = (i + 1 < P ? i + 1 : P - 1) / HW, c2 = (i + 2 < P ? i + 2 : P - 1) / HW,
                              c3 = (i + 3 < P ? i + 3 : P - 1) / HW;
                o.x = (o.x - bn_mean[c0]) * bn_invstd[c0];
                o.y = (o.y - bn_mean[c1]) * bn_invstd[c1];
                o.z = (o.z - bn_mean[c2]) * bn_invstd[c2];
                o.w = (o.w - bn_mean[c3]) * bn_invstd[c3];
            }
            if (VEC) {
                reinterpret_cast<float4 *>(yo)[g] = o;
            } else {
                const int64_t i = g * 4;
                yo[i] = o.x;
                if (i + 1 < P) yo[i + 1] = o.y;
                if (i + 2 < P) yo[i + 2] = o.z;
                if (i + 3 < P) yo[i + 3] = o.w;
            }
            so[g] = (uint8_t)((int)t0 | ((int)t1 << 1) | ((int)t2 << 2) | ((int)t3 << 3));
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void mfm_backward_kernel(const float *__restrict__ gy,
                                                              const uint8_t *__restrict__ sel,
                                                              const float *__restrict__ gscale, float *__restrict__ gx,
                                                              int64_t P, int64_t G, int64_t HW) {
    const int64_t n = blockIdx.y;
    const float *go = gy + n * P;
    const uint8_t *so = sel + n * G;
    float *ga = gx + n * 2 * P;
    float *gb = ga + P;
    const int64_t g0 = (int64_t)blockIdx.x * (kBlock * kGroupsPerThread) + threadIdx.x;
    float4 v[kGroupsPerThread];
    int bits[kGroupsPerThread];
#pragma unroll
    for (int k = 0; k < kGroupsPerThread; ++k) {
        const int64_t g = g0 + (int64_t)k * kBlock;
        if (g < G) {
            bits[k] = so[g];
            if (VEC) {
                v[k] = reinterpret_cast<const float4 *>(go)[g];
            } else {
                const int64_t i = g * 4;
                v[k].x = go[i];
                v[k].y = (i + 1 < P) ? go[i + 1] : 0.0f;
                v[k].z = (i + 2 < P) ? go[i + 2] : 0.0f;
                v[k].w = (i + 3 < P) ? go[i + 3] : 0.0f;
            }
            if (gscale) {  // backward of the BatchNorm that followed: gy * invstd[c]
                const int64_t i = g * 4;
                v[k].x *= gscale[i / HW];
                v[k].y *= gscale[(i + 1 < P ? i + 1 : P - 1) / HW];
                v[k].z *= gscale[(i + 2 < P ? i + 2 : P - 1) / HW];
                v[k].w *= gscale[(i + 3 < P ? i + 3 : P - 1) / HW];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kGroupsPerThread; ++k) {
        const int64_t g = g0 + (int64_t)k * kBlock;
        if (g < G) {
            const bool t0 = bits[k] & 1, t1 = bits[k] & 2, t2 = bits[k] & 4, t3 = bits[k] & 8;
            const float4 oa = make_float4(t0 ? 0.0f : v[k].x, t1 ? 0.0f : v[k].y, t2 ? 0.0f : v[k].z, t3 ? 0.0f : v[k].w);
            const float4 ob = make_float4(t0 ? v[k].x : 0.0f, t1 ? v[k].y : 0.0f, t2 ? v[k].z : 0.0f, t3 ? v[k].w : 0.0f);
            if (VEC) {
                reinterpret_cast<float4 *>(ga)[g] = oa;
                reinterpret_cast<float4 *>(gb)[g] = ob;
            } else {
                const int64_t i = g * 4;
                ga[i] = oa.x;
                gb[i] = ob.x;
                if (i + 1 < P) { ga[i + 1] = oa.y; gb[i + 1] = ob.y; }
                if (i + 2 < P) { ga[i + 2] = oa.z; gb[i + 2] = ob.z; }
                if (i + 3 < P) { ga[i + 3] = oa.w; gb[i + 3] = ob.w; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// MFM + MaxPool2d(2, 2)
// ---------------------------------------------------------------------------------------------------------

// One pooled output from its 2x2 window of (a, b) pairs, in ATen's order: MFM per position, then the pool scan
// (0,0), (0,1), (1,0), (1,1) with  take = (v > best || isnan(v)),  best = -inf initially.
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

// VEC path (W % 4 == 0, 16-byte aligned planes): thread = (c, ho, wq) -> two pooled outputs from four float4 loads.
__global__ __launch_bounds__(kBlock) void mfm_pool2_forward_vec_kernel(const float *__restrict__ x,
                                                                       const float *__restrict__ bias,
                                                                       const float *__restrict__ bn_mean,
                                                                       const float *__restrict__ bn_invstd,
                                                                       float *__restrict__ y,
                                                                       uint8_t *__restrict__ idx, int C, int H, int W) {
    const int Ho = H >> 1, W4 = W >> 2, Wo = W >> 1;
    const int64_t n = blockIdx.y;
    const int64_t items = (int64_t)C * Ho * W4;
    const int64_t plane = (int64_t)H * W;
    const float *xn = x + n * 2 * C * plane;
    float *yn = y + n * (int64_t)C * Ho * Wo;
    uint8_t *in = idx + n * (int64_t)C * Ho * Wo;
#pragma unroll 2
    for (int r = 0; r < 2; ++r) {
        const int64_t i = ((int64_t)blockIdx.x * 2 + r) * kBlock + threadIdx.x;
        if (i >= items) return;
        const int wq = (int)(i % W4);
        const int64_t t = i / W4;
        const int ho = (int)(t % Ho);
        const int c = (int)(t / Ho);
        const float *pa = xn + c * plane + (int64_t)(2 * ho) * W + 4 * wq;
        const float *pb = pa + (int64_t)C * plane;
        float4 a0 = *reinterpret_cast<const float4 *>(pa), a1 = *reinterpret_cast<const float4 *>(pa + W);
        float4 b0 = *reinterpret_cast<const float4 *>(pb), b1 = *reinterpret_cast<const float4 *>(pb + W);
        if (bias) {
            const float ba = bias[c], bb = bias[c + C];
            a0.x += ba; a0.y += ba; a0.z += ba; a0.w += ba; a1.x += ba; a1.y += ba; a1.z += ba; a1.w += ba;
            b0.x += bb; b0.y += bb; b0.z += bb; b0.w += bb; b1.x += bb; b1.y += bb; b1.z += bb; b1.w += bb;
        }
        int c0, c1;
        float2 o;
        o.x = pool_select(a0.x, b0.x, a0.y, b0.y, a1.x, b1.x, a1.y, b1.y, c0);
        o.y = pool_select(a0.z, b0.z, a0.w, b0.w, a1.z, b1.z, a1.w, b1.w, c1);
        if (bn_mean) {
            const float mu = bn_mean[c], is = bn_invstd[c];
            o.x = (o.x - mu) * is;
            o.y = (o.y - mu) * is;
        }
        reinterpret_cast<float2 *>(yn)[i] = o;              // (c, ho, 2wq..2wq+1) is item i of the flattened output pairs
        reinterpret_cast<uchar2 *>(in)[i] = make_uchar2((unsigned char)c0, (unsigned char)c1);
    }
}

// generic path: thread = one pooled output
__global__ __launch_bounds__(kBlock) void mfm_pool2_forward_scalar_kernel(const float *__restrict__ x,
                                                                          const float *__restrict__ bias,
                                                                          const float *__restrict__ bn_mean,
                                                                          const float *__restrict__ bn_invstd,
                                                                          float *__restrict__ y,
                                                                          uint8_t *__restrict__ idx, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t n = blockIdx.y;
    const int64_t items = (int64_t)C * Ho * Wo;
    const int64_t plane = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= items) return;
    const int wo = (int)(i % Wo);
    const int64_t t = i / Wo;
    const int ho = (int)(t % Ho);
    const int c = (int)(t / Ho);
    const float *pa = x + n * 2 * C * plane + c * plane + (int64_t)(2 * ho) * W + 2 * wo;
    const float *pb = pa + (int64_t)C * plane;
    int code;
    const float ba = bias ? bias[c] : 0.0f, bb = bias ? bias[c + C] : 0.0f;
    float v;
    if (bias)
        v = pool_select(pa[0] + ba, pb[0] + bb, pa[1] + ba, pb[1] + bb, pa[W] + ba, pb[W] + bb, pa[W + 1] + ba,
                        pb[W + 1] + bb, code);
    else
        v = pool_select(pa[0], pb[0], pa[1], pb[1], pa[W], pb[W], pa[W + 1], pb[W + 1], code);
    if (bn_mean) v = (v - bn_mean[c]) * bn_invstd[c];
    y[n * items + i] = v;
    idx[n * items + i] = (uint8_t)code;
}

// VEC backward: thread = (c, ho, wq): reads 2 pooled gradients + 2 codes, writes the 2x4 window of both halves.
// When H is odd, the threads of the last pooled row also zero the trailing input row.
__global__ __launch_bounds__(kBlock) void mfm_pool2_backward_vec_kernel(const float *__restrict__ gy,
                                                                        const uint8_t *__restrict__ idx,
                                                                        const float *__restrict__ gscale,
                                                                        float *__restrict__ gx, int C, int H, int W) {
    const int Ho = H >> 1, W4 = W >> 2, Wo = W >> 1;
    const int64_t n = blockIdx.y;
    const int64_t items = (int64_t)C * Ho * W4;
    const int64_t plane = (int64_t)H * W;
    const float *gn = gy + n * (int64_t)C * Ho * Wo;
    const uint8_t *in = idx + n * (int64_t)C * Ho * Wo;
    float *xn = gx + n * 2 * C * plane;
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 2
    for (int r = 0; r < 2; ++r) {
        const int64_t i = ((int64_t)blockIdx.x * 2 + r) * kBlock + threadIdx.x;
        if (i >= items) return;
        const int wq = (int)(i % W4);
        const int64_t t = i / W4;
        const int ho = (int)(t % Ho);
        const int c = (int)(t / Ho);
        float2 g = reinterpret_cast<const float2 *>(gn)[i];
        if (gscale) {
            g.x *= gscale[c];
            g.y *= gscale[c];
        }
        const uchar2 code = reinterpret_cast<const uchar2 *>(in)[i];
        // window of output 0 = columns 0-1, of output 1 = columns 2-3; rows dh = 0, 1; halves a / b
        const int p0 = code.x & 3, p1 = code.y & 3;
        const bool h0 = code.x & 4, h1 = code.y & 4;
        const float a0x = (!h0 && p0 == 0) ? g.x : 0.0f, a0y = (!h0 && p0 == 1) ? g.x : 0.0f;
        const float a1x = (!h0 && p0 == 2) ? g.x : 0.0f, a1y = (!h0 && p0 == 3) ? g.x : 0.0f;
        const float b0x = (h0 && p0 == 0) ? g.x : 0.0f, b0y = (h0 && p0 == 1) ? g.x : 0.0f;
        const float b1x = (h0 && p0 == 2) ? g.x : 0.0f, b1y = (h0 && p0 == 3) ? g.x : 0.0f;
        const float a0z = (!h1 && p1 == 0) ? g.y : 0.0f, a0w = (!h1 && p1 == 1) ? g.y : 0.0f;
        const float a1z = (!h1 && p1 == 2) ? g.y : 0.0f, a1w = (!h1 && p1 == 3) ? g.y : 0.0f;
        const float b0z = (h1 && p1 == 0) ? g.y : 0.0f, b0w = (h1 && p1 == 1) ? g.y : 0.0f;
        const float b1z = (h1 && p1 == 2) ? g.y : 0.0f, b1w = (h1 && p1 == 3) ? g.y : 0.0f;
        float *pa = xn + c * plane + (int64_t)(2 * ho) * W + 4 * wq;
        float *pb = pa + (int64_t)C * plane;
        *reinterpret_cast<float4 *>(pa) = make_float4(a0x, a0y, a0z, a0w);
        *reinterpret_cast<float4 *>(pa + W) = make_float4(a1x, a1y, a1z, a1w);
        *reinterpret_cast<float4 *>(pb) = make_float4(b0x, b0y, b0z, b0w);
        *reinterpret_cast<float4 *>(pb + W) = make_float4(b1x, b1y, b1z, b1w);
        if ((H & 1) && ho == Ho - 1) {
            *reinterpret_cast<float4 *>(pa + 2 * W) = zero;
            *reinterpret_cast<float4 *>(pb + 2 * W) = zero;
        }
    }
}

// generic backward: gx was zero-filled by the host side; thread = one pooled output scatters its gradient
__global__ __launch_bounds__(kBlock) void mfm_pool2_backward_scalar_kernel(const float *__restrict__ gy,
                                                                           const uint8_t *__restrict__ idx,
                                                                           const float *__restrict__ gscale,
                                                                           float *__restrict__ gx, int C, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t n = blockIdx.y;
    const int64_t items = (int64_t)C * Ho * Wo;
    const int64_t plane = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= items) return;
    const int wo = (int)(i % Wo);
    const int64_t t = i / Wo;
    const int ho = (int)(t % Ho);
    const int c = (int)(t / Ho);
    const int code = idx[n * items + i];
    const int64_t half = (code & 4) ? (int64_t)C * plane : 0;
    gx[n * 2 * C * plane + half + c * plane + (int64_t)(2 * ho + ((code >> 1) & 1)) * W + 2 * wo + (code & 1)] =
        gscale ? gy[n * items + i] * gscale[c] : gy[n * items + i];
}

constexpr int64_t kMaxGridY = 65535;

}  // namespace

#define LCNN_REQUIRE(cond) \
    do {                   \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

size_t advstep_mfm_sel_bytes(int64_t N, int64_t C, int64_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return (size_t)N * (size_t)ceil_div(C * HW, 4);
}

int advstep_mfm_forward_f32(const float *x, const float *bias, const float *bn_mean, const float *bn_invstd, float *y,
                            uint8_t *sel, int64_t N, int64_t C, int64_t HW, advstep_stream_t stream) {
    LCNN_REQUIRE(N >= 0 && C >= 0 && HW >= 0);
    if (N == 0 || C == 0 || HW == 0) return ADVSTEP_OK;
    LCNN_REQUIRE(x && y && sel && N <= kMaxGridY);
    const int64_t P = C * HW, G = ceil_div(P, 4);
    const dim3 grid((unsigned)ceil_div(G, kBlock * kGroupsPerThread), (unsigned)N);
    if (P % 4 == 0 && aligned16(x) && aligned16(y))
        hipLaunchKernelGGL(mfm_forward_kernel<true>, grid, dim3(kBlock), 0, as_stream(stream), x, bias, bn_mean, bn_invstd, y, sel, P, G, HW, C);
    else
        hipLaunchKernelGGL(mfm_forward_kernel<false>, grid, dim3(kBlock), 0, as_stream(stream), x, bias, bn_mean, bn_invstd, y, sel, P, G, HW, C);
    return status_after_launch();
}

int advstep_mfm_backward_f32(const float *gy, const uint8_t *sel, const float *gscale, float *gx, int64_t N, int64_t C,
                             int64_t HW, advstep_stream_t stream) {
    LCNN_REQUIRE(N >= 0 && C >= 0 && HW >= 0);
    if (N == 0 || C == 0 || HW == 0) return ADVSTEP_OK;
    LCNN_REQUIRE(gy && sel && gx && N <= kMaxGridY);
    const int64_t P = C * HW, G = ceil_div(P, 4);
    const dim3 grid((unsigned)ceil_div(G, kBlock * kGroupsPerThread), (unsigned)N);
    if (P % 4 == 0 && aligned16(gy) && aligned16(gx))
        hipLaunchKernelGGL(mfm_backward_kernel<true>, grid, dim3(kBlock), 0, as_stream(stream), gy, sel, gscale, gx, P, G, HW);
    else
        hipLaunchKernelGGL(mfm_backward_kernel<false>, grid, dim3(kBlock), 0, as_stream(stream), gy, sel, gscale, gx, P, G, HW);
    return status_after_launch();
}

int advstep_mfm_pool2_forward_f32(const float *x, const float *bias, const float *bn_mean, const float *bn_invstd,
                                  float *y, uint8_t *idx, int64_t N, int64_t C, int64_t H, int64_t W,
                                  advstep_stream_t stream) {
    LCNN_REQUIRE(N >= 0 && C >= 0 && H >= 0 && W >= 0);
    const int64_t Ho = H / 2, Wo = W / 2;
    if (N == 0 || C == 0 || Ho == 0 || Wo == 0) return ADVSTEP_OK;
    LCNN_REQUIRE(x && y && idx && N <= kMaxGridY && C <= INT32_MAX && H <= INT32_MAX && W <= INT32_MAX);
    hipStream_t st = as_stream(stream);
    if (W % 4 == 0 && aligned16(x) && ((reinterpret_cast<uintptr_t>(y) & 7u) == 0) &&
        ((reinterpret_cast<uintptr_t>(idx) & 1u) == 0)) {
        const int64_t items = C * Ho * (W / 4);
        const dim3 grid((unsigned)ceil_div(items, 2 * kBlock), (unsigned)N);
        hipLaunchKernelGGL(mfm_pool2_forward_vec_kernel, grid, dim3(kBlock), 0, st, x, bias, bn_mean, bn_invstd, y, idx, (int)C, (int)H, (int)W);
    } else {
        const dim3 grid((unsigned)ceil_div(C * Ho * Wo, kBlock), (unsigned)N);
        hipLaunchKernelGGL(mfm_pool2_forward_scalar_kernel, grid, dim3(kBlock), 0, st, x, bias, bn_mean, bn_invstd, y, idx, (int)C, (int)H, (int)W);
    }
    return status_after_launch();
}

int advstep_mfm_pool2_backward_f32(const float *gy, const uint8_t *idx, const float *gscale, float *gx, int64_t N,
                                   int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    LCNN_REQUIRE(N >= 0 && C >= 0 && H >= 0 && W >= 0);
    if (N == 0 || C == 0 || H == 0 || W == 0) return ADVSTEP_OK;
    LCNN_REQUIRE(gx && N <= kMaxGridY && C <= INT32_MAX && H <= INT32_MAX && W <= INT32_MAX);
    hipStream_t st = as_stream(stream);
    const int64_t Ho = H / 2, Wo = W / 2;
    if (Ho == 0 || Wo == 0) {  // nothing was pooled: the whole gradient is zero
        return hipMemsetAsync(gx, 0, (size_t)N * 2 * C * H * W * sizeof(float), st) == hipSuccess ? ADVSTEP_OK
                                                                                                 : ADVSTEP_ELAUNCH;
    }
    LCNN_REQUIRE(gy && idx);
    if (W % 4 == 0 && aligned16(gx) && ((reinterpret_cast<uintptr_t>(gy) & 7u) == 0) &&
        ((reinterpret_cast<uintptr_t>(idx) & 1u) == 0)) {
        const int64_t items = C * Ho * (W / 4);
        const dim3 grid((unsigned)ceil_div(items, 2 * kBlock), (unsigned)N);
        hipLaunchKernelGGL(mfm_pool2_backward_vec_kernel, grid, dim3(kBlock), 0, st, gy, idx, gscale, gx, (int)C, (int)H, (int)W);
    } else {
        if (hipMemsetAsync(gx, 0, (size_t)N * 2 * C * H * W * sizeof(float), st) != hipSuccess) return ADVSTEP_ELAUNCH;
        const dim3 grid((unsigned)ceil_div(C * Ho * Wo, kBlock), (unsigned)N);
        hipLaunchKernelGGL(mfm_pool2_backward_scalar_kernel, grid, dim3(kBlock), 0, st, gy, idx, gscale, gx, (int)C, (int)H, (int)W);
    }
    return status_after_launch();
}

}  // extern "C"
