// detector_elem.hip — fused elementwise / pooling kernels of the SpecRNet and RawNet3 detectors (C ABI:
// include/advstep_detector.h).  All HBM-bound: 16 B per lane, one pass per tensor, grid (N * C, tiles) so the channel of a
// workgroup is uniform (per-channel constants are scalar loads).  Planes (n, c) ride on grid.x, tiles of a plane on grid.y.
//
// Replaced ATen / MIOpen chains (each op a full read + write of the activation, at B = 128 up to 331 MB per tensor):
//   conv bias add_ -> MIOpenBatchNormFwdInfer -> leaky_relu            =>  affine_act_forward (mode 0)
//   leaky_relu_backward -> batch_norm_elementwise_backward_eval         =>  affine_act_backward
//   conv bias add_ -> relu -> batch_norm (RawNet3)                      =>  affine_act_forward (mode 1)
//   2 x conv bias add_ -> add -> max_pool_forward (values + int64 idx)  =>  add_maxpool2_forward (values + 1 byte)
//   max_pool_backward_nchw (98 us average in the SpecRNet step)         =>  maxpool2_backward
//   mul -> add -> max_pool_forward and their backward                   =>  gate_maxpool2_forward / _backward

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_detector.h"

namespace {

constexpr int kBlock = 256;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- per-channel affine + activation: grid (N * C planes, tiles of 4 x 256 float4 per plane) -------------------------------
constexpr int kVecPerThread = 4;

constexpr float kSeluAlpha = 1.6732632423543772848170429916717f, kSeluScale = 1.0507009873554804934193349852946f;

template <int MODE>
__device__ __forceinline__ float act_fwd(float x, float s, float t, float pre, float slope) {
    if (MODE == 0) {
        const float v = x * s + t;
        return v > 0.0f ? v : v * slope;
    }
    if (MODE == 2) {                                              // at::selu = elu(x, alpha, scale): exp(v) - 1, not expm1
        const float v = x * s + t;
        return v <= 0.0f ? (expf(v) - 1.0f) * (kSeluAlpha * kSeluScale) : v * kSeluScale;
    }
    const float r = x + pre;
    return (r > 0.0f ? r : (r != r ? r : 0.0f)) * s + t;      // at::relu = clamp_min(0): NaN stays NaN
}
template <int MODE>
__device__ __forceinline__ float act_bwd(float gy, float x, float s, float t, float pre, float slope) {
    if (MODE == 0) return gy * ((x * s + t) > 0.0f ? 1.0f : slope) * s;   // leaky_relu_backward: x > 0 ? g : g * slope
    if (MODE == 2) {                                                        // elu_backward from the input
        const float v = x * s + t;
        return gy * (v <= 0.0f ? (kSeluAlpha * kSeluScale) * expf(v) : kSeluScale) * s;
    }
    return (x + pre) <= 0.0f ? 0.0f : gy * s;                               // threshold_backward: x <= 0 ? 0 : g (NaN: g)
}

// VEC: every tensor base is 16-byte aligned.  A plane (n, c) starts at element nc * P, which is 16-byte aligned only when
// P % 4 == 0 (SpecRNet's 2-D planes) — RawNet3's planes are 6435 / 1287 / 429 frames long — so a plane is split into a
// scalar head (up to its first aligned element), a float4 body and a scalar tail; the head and tail (<= 6 elements) are
// handled by the first threads of the plane's first workgroup.
template <int MODE, bool BWD, bool VEC>
__global__ __launch_bounds__(kBlock) void affine_act_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                                            const float *__restrict__ scale, const float *__restrict__ shift,
                                                            const float *__restrict__ pre, float *__restrict__ out, int64_t C,
                                                            int64_t P, float slope) {
    const int64_t nc = blockIdx.x;       // plane on x (up to 2^31 - 1 of them), tile on y
    const int c = (int)(nc % C);
    const float s = scale[c], t = shift[c], pr = pre ? pre[c] : 0.0f;
    const float *xp = x + nc * P;
    const float *gp = BWD ? gy + nc * P : nullptr;
    float *op = out + nc * P;
    if (VEC) {
        int64_t head = (4 - ((nc * P) & 3)) & 3;
        if (head > P) head = P;
        const int64_t G = (P - head) / 4, tail0 = head + 4 * G;
        const float4 *xv4 = reinterpret_cast<const float4 *>(xp + head);
        const float4 *gv4 = BWD ? reinterpret_cast<const float4 *>(gp + head) : nullptr;
        float4 *ov4 = reinterpret_cast<float4 *>(op + head);
        const int64_t g0 = (int64_t)blockIdx.y * (kBlock * kVecPerThread) + threadIdx.x;
        float4 xv[kVecPerThread], gv[kVecPerThread];
#pragma unroll
        for (int k = 0; k < kVecPerThread; ++k) {
            const int64_t g = g0 + (int64_t)k * kBlock;
            if (g < G) {
                xv[k] = xv4[g];
                if (BWD) gv[k] = gv4[g];
            }
        }
#pragma unroll
        for (int k = 0; k < kVecPerThread; ++k) {
            const int64_t g = g0 + (int64_t)k * kBlock;
            if (g < G) {
                float4 o;
                if (BWD) {
                    o.x = act_bwd<MODE>(gv[k].x, xv[k].x, s, t, pr, slope);
                    o.y = act_bwd<MODE>(gv[k].y, xv[k].y, s, t, pr, slope);
                    o.z = act_bwd<MODE>(gv[k].z, xv[k].z, s, t, pr, slope);
                    o.w = act_bwd<MODE>(gv[k].w, xv[k].w, s, t, pr, slope);
                } else {
                    o.x = act_fwd<MODE>(xv[k].x, s, t, pr, slope);
                    o.y = act_fwd<MODE>(xv[k].y, s, t, pr, slope);
                    o.z = act_fwd<MODE>(xv[k].z, s, t, pr, slope);
                    o.w = act_fwd<MODE>(xv[k].w, s, t, pr, slope);
                }
                ov4[g] = o;
            }
        }
        if (blockIdx.y == 0 && threadIdx.x < 8) {
            // threads 0-3: head element threadIdx.x; threads 4-7: tail element tail0 + threadIdx.x - 4
            const int64_t i = threadIdx.x < 4 ? (int64_t)threadIdx.x : tail0 + threadIdx.x - 4;
            const bool live = threadIdx.x < 4 ? i < head : i < P;
            if (live) op[i] = BWD ? act_bwd<MODE>(gp[i], xp[i], s, t, pr, slope) : act_fwd<MODE>(xp[i], s, t, pr, slope);
        }
    } else {
        const int64_t i0 = ((int64_t)blockIdx.y * (kBlock * kVecPerThread) + threadIdx.x) * 4;
#pragma unroll
        for (int k = 0; k < kVecPerThread; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t i = i0 + (int64_t)k * kBlock * 4 + e;
                if (i < P) op[i] = BWD ? act_bwd<MODE>(gp[i], xp[i], s, t, pr, slope) : act_fwd<MODE>(xp[i], s, t, pr, slope);
            }
    }
}

template <bool BWD>
int launch_affine(const float *gy, const float *x, const float *scale, const float *shift, const float *pre, float *out,
                  int64_t N, int64_t C, int64_t P, int mode, float slope, hipStream_t st) {
    if (N < 0 || C < 0 || P < 0 || (mode != 0 && mode != 1 && mode != 2)) return ADVSTEP_EINVAL;
    if (N * C * P == 0) return ADVSTEP_OK;
    const int64_t tiles = ceil_div(ceil_div(P, 4), kBlock * kVecPerThread);
    if (!x || !scale || !shift || !out || (BWD && !gy) || N * C > 0x7fffffffLL || tiles > 65535) return ADVSTEP_EINVAL;
    const bool vec = aligned16(x) && aligned16(out) && (!BWD || aligned16(gy));
    const dim3 grid((unsigned)(N * C), (unsigned)tiles), block(kBlock);
#define GO(MODE, VEC)                                                                                                   \
    hipLaunchKernelGGL((affine_act_kernel<MODE, BWD, VEC>), grid, block, 0, st, gy, x, scale, shift, pre, out, C, P, slope)
    if (mode == 0) { if (vec) GO(0, true); else GO(0, false); }
    else if (mode == 1) { if (vec) GO(1, true); else GO(1, false); }
    else { if (vec) GO(2, true); else GO(2, false); }
#undef GO
    return status_after_launch();
}

// ---- one link of RawNet3's Res2Net chain (src/models/rawnet3.py:244-258) ------------------------------------------------
// forward : y = relu(h + pre[c]) * scale[c] + shift[c]  (the branch's `bn(relu(conv(.)))`), written into its channel slice of
//           the concatenated tensor (batch stride y_bs), and z = y + other  — the NEXT branch's input `sp + spx[i + 1]`
//           (other = the next group, a channel slice with batch stride o_bs) — in the same pass.
// backward: gx = (h + pre <= 0) ? 0 : (g1 + g2) * scale  where g1 is the slice of d(concatenated) (batch stride g1_bs) and
//           g2 the gradient arriving from the next branch's input (batch stride g2_bs); either extra operand may be absent.
// h, z and gx are contiguous (N, C, P).  Plane (n, c) of a strided operand starts at n * bs + c * P; with every batch stride a
// multiple of 4 and 16-byte aligned bases all operands of a plane share one alignment phase ((c * P) & 3), so the scalar head /
// float4 body / scalar tail split of affine_act_kernel applies (VEC); otherwise element by element.
template <bool BWD, bool VEC>
__global__ __launch_bounds__(kBlock) void res2net_link_kernel(const float *__restrict__ h, const float *__restrict__ a,
                                                              int64_t a_bs, const float *__restrict__ b, int64_t b_bs,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              const float *__restrict__ pre, float *__restrict__ o1,
                                                              int64_t o1_bs, float *__restrict__ o2, int64_t C, int64_t P) {
    // FWD: a = other (may be null), b unused; o1 = y (strided), o2 = z (contiguous, null when a is null)
    // BWD: a = g1 (strided), b = g2 (may be null); o1 = gx (contiguous: o1_bs = C * P), o2 unused
    const int64_t nc = blockIdx.x, n = nc / C;
    const int c = (int)(nc - n * C);
    const float s = scale[c], t = shift[c], pr = pre ? pre[c] : 0.0f;
    const float *hp = h + nc * P;
    const float *ap = a ? a + n * a_bs + (int64_t)c * P : nullptr;
    const float *bp = (BWD && b) ? b + n * b_bs + (int64_t)c * P : nullptr;
    float *o1p = o1 + n * o1_bs + (int64_t)c * P;
    float *o2p = (!BWD && o2) ? o2 + nc * P : nullptr;
    auto one = [&](int64_t i) {
        if (BWD) {
            const float g = ap[i] + (bp ? bp[i] : 0.0f);
            o1p[i] = act_bwd<1>(g, hp[i], s, t, pr, 0.0f);
        } else {
            const float y = act_fwd<1>(hp[i], s, t, pr, 0.0f);
            o1p[i] = y;
            if (o2p) o2p[i] = y + ap[i];
        }
    };
    if (VEC) {
        int64_t head = (4 - (((int64_t)c * P) & 3)) & 3;
        if (head > P) head = P;
        const int64_t G = (P - head) / 4, tail0 = head + 4 * G;
        const int64_t g0 = (int64_t)blockIdx.y * (kBlock * kVecPerThread) + threadIdx.x;
        float4 hv[kVecPerThread], av[kVecPerThread], bv[kVecPerThread];
#pragma unroll
        for (int k = 0; k < kVecPerThread; ++k) {
            const int64_t g = g0 + (int64_t)k * kBlock;
            if (g < G) {
                hv[k] = reinterpret_cast<const float4 *>(hp + head)[g];
                av[k] = ap ? reinterpret_cast<const float4 *>(ap + head)[g] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                bv[k] = bp ? reinterpret_cast<const float4 *>(bp + head)[g] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int k = 0; k < kVecPerThread; ++k) {
            const int64_t g = g0 + (int64_t)k * kBlock;
            if (g < G) {
                float4 r;
                if (BWD) {
                    r.x = act_bwd<1>(av[k].x + bv[k].x, hv[k].x, s, t, pr, 0.0f);
                    r.y = act_bwd<1>(av[k].y + bv[k].y, hv[k].y, s, t, pr, 0.0f);
                    r.z = act_bwd<1>(av[k].z + bv[k].z, hv[k].z, s, t, pr, 0.0f);
                    r.w = act_bwd<1>(av[k].w + bv[k].w, hv[k].w, s, t, pr, 0.0f);
                    reinterpret_cast<float4 *>(o1p + head)[g] = r;
                } else {
                    r.x = act_fwd<1>(hv[k].x, s, t, pr, 0.0f);
                    r.y = act_fwd<1>(hv[k].y, s, t, pr, 0.0f);
                    r.z = act_fwd<1>(hv[k].z, s, t, pr, 0.0f);
                    r.w = act_fwd<1>(hv[k].w, s, t, pr, 0.0f);
                    reinterpret_cast<float4 *>(o1p + head)[g] = r;
                    if (o2p)
                        reinterpret_cast<float4 *>(o2p + head)[g] =
                            make_float4(r.x + av[k].x, r.y + av[k].y, r.z + av[k].z, r.w + av[k].w);
                }
            }
        }
        if (blockIdx.y == 0 && threadIdx.x < 8) {
            const int64_t i = threadIdx.x < 4 ? (int64_t)threadIdx.x : tail0 + threadIdx.x - 4;
            if (threadIdx.x < 4 ? i < head : i < P) one(i);
        }
    } else {
        const int64_t i0 = ((int64_t)blockIdx.y * (kBlock * kVecPerThread) + threadIdx.x) * 4;
#pragma unroll
        for (int k = 0; k < kVecPerThread; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t i = i0 + (int64_t)k * kBlock * 4 + e;
                if (i < P) one(i);
            }
    }
}

template <bool BWD>
int launch_link(const float *h, const float *a, int64_t a_bs, const float *b, int64_t b_bs, const float *scale, const float *shift,
                const float *pre, float *o1, int64_t o1_bs, float *o2, int64_t N, int64_t C, int64_t P, hipStream_t st) {
    const int64_t tiles = ceil_div(ceil_div(P, 4), kBlock * kVecPerThread);
    if (N * C > 0x7fffffffLL || tiles > 65535) return ADVSTEP_EINVAL;
    const bool vec = aligned16(h) && aligned16(o1) && (!a || (aligned16(a) && a_bs % 4 == 0)) && (!b || (aligned16(b) && b_bs % 4 == 0)) &&
                     o1_bs % 4 == 0 && (!o2 || aligned16(o2)) && (C * P) % 4 == 0;
    const dim3 grid((unsigned)(N * C), (unsigned)tiles), block(kBlock);
    if (vec)
        hipLaunchKernelGGL((res2net_link_kernel<BWD, true>), grid, block, 0, st, h, a, a_bs, b, b_bs, scale, shift, pre, o1, o1_bs, o2, C, P);
    else
        hipLaunchKernelGGL((res2net_link_kernel<BWD, false>), grid, block, 0, st, h, a, a_bs, b, b_bs, scale, shift, pre, o1, o1_bs, o2, C, P);
    return status_after_launch();
}

// ---- 2x2 max pooling family: thread = 2 horizontally adjacent pooled outputs -------------------------------------------
// at::native::max_pool_forward_nchw: scan the window row-major, take val when (val > max) || isnan(val); start max = -inf.
__device__ __forceinline__ float pool4(float v00, float v01, float v10, float v11, int &code) {
    float best = -INFINITY;
    code = 0;
    if (v00 > best || v00 != v00) { best = v00; code = 0; }
    if (v01 > best || v01 != v01) { best = v01; code = 1; }
    if (v10 > best || v10 != v10) { best = v10; code = 2; }
    if (v11 > best || v11 != v11) { best = v11; code = 3; }
    return best;
}

// KIND 0: a (+ b) (+ bias[c]);  KIND 1: x * gate[n, c] + gate[n, c]
template <int KIND>
__global__ __launch_bounds__(kBlock) void pool2_forward_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                               const float *__restrict__ bias, const float *__restrict__ gate,
                                                               float *__restrict__ y, uint8_t *__restrict__ sel, int64_t C,
                                                               int H, int W, float *__restrict__ xw) {
    const int64_t nc = blockIdx.x;
    const int Ho = H >> 1, Wo = W >> 1, Wp = (Wo + 1) >> 1;      // Wp pairs of pooled outputs per row
    const int64_t idx = (int64_t)blockIdx.y * kBlock + threadIdx.x;
    if (idx >= (int64_t)Ho * Wp) return;
    const int i = (int)(idx / Wp), jp = (int)(idx - (int64_t)i * Wp), j0 = 2 * jp;
    const float add = KIND == 0 ? (bias ? bias[nc % C] : 0.0f) : gate[nc];
    const float mul = KIND == 1 ? gate[nc] : 1.0f;
    const float *ap = a + (nc * H + 2 * i) * W + 2 * j0;
    const float *bp = (KIND == 0 && b) ? b + (nc * H + 2 * i) * W + 2 * j0 : nullptr;
    const bool two = j0 + 1 < Wo;
    float r0[4], r1[4];
    if (two && (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | (bp ? reinterpret_cast<uintptr_t>(b) : 0)) & 15u) == 0) {
        float4 t0 = *reinterpret_cast<const float4 *>(ap), t1 = *reinterpret_cast<const float4 *>(ap + W);
        r0[0] = t0.x, r0[1] = t0.y, r0[2] = t0.z, r0[3] = t0.w;
        r1[0] = t1.x, r1[1] = t1.y, r1[2] = t1.z, r1[3] = t1.w;
        if (bp) {
            t0 = *reinterpret_cast<const float4 *>(bp), t1 = *reinterpret_cast<const float4 *>(bp + W);
            r0[0] += t0.x, r0[1] += t0.y, r0[2] += t0.z, r0[3] += t0.w;
            r1[0] += t1.x, r1[1] += t1.y, r1[2] += t1.z, r1[3] += t1.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = e < 2 || two;
            r0[e] = in ? ap[e] + (bp ? bp[e] : 0.0f) : 0.0f;
            r1[e] = in ? ap[W + e] + (bp ? bp[W + e] : 0.0f) : 0.0f;
        }
    }
    float raw[KIND == 1 ? 8 : 1];                                 // KIND 1: the un-gated inputs, for `xw`
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (KIND == 1) { raw[e] = r0[e]; raw[4 + e] = r1[e]; }
        if (KIND == 0) { r0[e] += add; r1[e] += add; }
        else { r0[e] = r0[e] * mul + add; r1[e] = r1[e] * mul + add; }
    }
    int c0, c1;
    const float v0 = pool4(r0[0], r0[1], r1[0], r1[1], c0), v1 = pool4(r0[2], r0[3], r1[2], r1[3], c1);
    const int64_t o = (nc * Ho + i) * Wo + j0;
    y[o] = v0;
    sel[o] = (uint8_t)c0;
    if (two) {
        y[o + 1] = v1;
        sel[o + 1] = (uint8_t)c1;
    }
    if (KIND == 1 && xw) {
        // the winner's un-gated input: all the gate's gradient needs of x (sum gy * (x_winner + 1)), at a quarter of x's size
        const float w0 = c0 == 0 ? raw[0] : c0 == 1 ? raw[1] : c0 == 2 ? raw[4] : raw[5];
        const float w1 = c1 == 0 ? raw[2] : c1 == 1 ? raw[3] : c1 == 2 ? raw[6] : raw[7];
        xw[o] = w0;
        if (two) xw[o + 1] = w1;
    }
}

// ggate[row] = sum_j gy[row][j] * (xw[row][j] + 1) over one pooled plane per workgroup, fixed order (thread-strided partial sums,
// wave butterfly, the 4 wave sums in index order)
__global__ __launch_bounds__(kBlock) void gate_grad_pooled_kernel(const float *__restrict__ gy, const float *__restrict__ xw,
                                                                  float *__restrict__ ggate, int64_t L) {
    const int64_t row = blockIdx.x;
    const float *a = gy + row * L, *b = xw + row * L;
    float acc = 0.0f;
    for (int64_t j = threadIdx.x; j < L; j += kBlock) acc += a[j] * (b[j] + 1.0f);
    __shared__ float ws[kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ggate[row] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}

// thread = one input row segment of 4 floats (2 pooled outputs); writes the full-resolution gradient incl. odd tails
// g == nullptr: only the gate's partial sums (GATE); addc (per plane, may be null): a constant added to EVERY element of the
// plane — the share of d x that arrives through the gate's `mean_t x` (SpecRNet's attention: detector_ops._AttendPool).
template <bool GATE>
__global__ __launch_bounds__(kBlock) void pool2_backward_kernel(const float *__restrict__ gy, const uint8_t *__restrict__ sel,
                                                                const float *__restrict__ x, const float *__restrict__ gate,
                                                                float *__restrict__ g, float *__restrict__ partial, int H,
                                                                int W, int blocks, const float *__restrict__ addc) {
    const int64_t nc = blockIdx.x;
    const int Ho = H >> 1, Wo = W >> 1;
    const int Wq = (W + 3) >> 2;                                  // 4-float segments per input row
    const int64_t idx = (int64_t)blockIdx.y * kBlock + threadIdx.x;
    float acc = 0.0f;
    if (idx < (int64_t)H * Wq) {
        const int h = (int)(idx / Wq), q = (int)(idx - (int64_t)h * Wq), w0 = 4 * q;
        const int i = h >> 1, dh = h & 1;
        const float gt = GATE ? gate[nc] : 1.0f;
        const float c0 = addc ? addc[nc] : 0.0f;
        float out[4] = {c0, c0, c0, c0};
        if (i < Ho) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * q + e;
                if (j < Wo) {
                    const int64_t o = (nc * Ho + i) * Wo + j;
                    const int code = sel[o];
                    if ((code >> 1) == dh) {
                        const float gv = gy[o];
                        out[2 * e + (code & 1)] = gv * gt + c0;
                        if (GATE && partial) acc += gv * (x[(nc * H + h) * W + w0 + 2 * e + (code & 1)] + 1.0f);
                    }
                }
            }
        }
        if (g) {
            float *gp = g + (nc * H + h) * W + w0;
            if (w0 + 3 < W && (W & 3) == 0) *reinterpret_cast<float4 *>(gp) = make_float4(out[0], out[1], out[2], out[3]);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (w0 + e < W) gp[e] = out[e];
        }
    }
    if (GATE && partial) {
        // fixed-order workgroup sum: wave butterfly, then the 4 wave sums in index order
        __shared__ float ws[kBlock / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[nc * blocks + blockIdx.y] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
    }
}


// ---- residual add + MaxPool1d(k) (kernel = stride = k, 2 <= k <= 8): thread = one pooled output ---------------------------
// at::native max-pool scan rule as pool4 above; sel = winner's offset inside its window (one byte).
__global__ __launch_bounds__(kBlock) void pool1d_forward_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                float *__restrict__ y, uint8_t *__restrict__ sel, int64_t L,
                                                                int64_t Lo, int k) {
    const int64_t nc = blockIdx.x;
    const int64_t j = (int64_t)blockIdx.y * kBlock + threadIdx.x;
    if (j >= Lo) return;
    const float *ap = a + nc * L + j * k;
    const float *bp = b ? b + nc * L + j * k : nullptr;
    float best = -INFINITY;
    int code = 0;
    for (int e = 0; e < k; ++e) {
        const float v = bp ? ap[e] + bp[e] : ap[e];
        if (v > best || v != v) { best = v; code = e; }
    }
    y[nc * Lo + j] = best;
    sel[nc * Lo + j] = (uint8_t)code;
}

// thread = one pooled output: writes its k-wide window of the gradient (and, for the last window, the dropped tail)
__global__ __launch_bounds__(kBlock) void pool1d_backward_kernel(const float *__restrict__ gy, const uint8_t *__restrict__ sel,
                                                                 float *__restrict__ g, int64_t L, int64_t Lo, int k) {
    const int64_t nc = blockIdx.x;
    const int64_t j = (int64_t)blockIdx.y * kBlock + threadIdx.x;
    if (j >= Lo) return;
    const float gv = gy[nc * Lo + j];
    const int code = sel[nc * Lo + j];
    float *gp = g + nc * L + j * k;
    for (int e = 0; e < k; ++e) gp[e] = e == code ? gv : 0.0f;
    if (j == Lo - 1)
        for (int64_t i = Lo * k; i < L; ++i) g[nc * L + i] = 0.0f;
}

// ---- the tail of a RawNet3 Bottle2neck in one pass each way (src/models/rawnet3.py:262-269) -----------------------------------
//     out = bn3(relu(conv3(.)));  out += residual;  out = MaxPool1d(k)(out)
// forward : y = pool_k(relu(h + pre[c]) * scale[c] + shift[c] + res), sel;  the activated tensor is never written.
// backward: g = unpool(gy, sel);  g_res = g;  g_h = (h + pre <= 0) ? 0 : g * scale   (the activation mask is recomputed from h).
// A workgroup owns kTailOut pooled outputs of one (n, c) row: their kTailOut * k inputs are read with consecutive lanes on
// consecutive elements and meet in LDS (window stride k = 3 or 5 is coprime to the 32 banks); the thread-per-window kernels
// above read lanes 4 k bytes apart (2.8 TB/s on these 1.7 GB tensors).
constexpr int kTailOut = 448;

__global__ __launch_bounds__(kBlock) void tail_pool1d_forward_kernel(const float *__restrict__ h, const float *__restrict__ res,
                                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                                     const float *__restrict__ pre, float *__restrict__ y,
                                                                     uint8_t *__restrict__ sel, int64_t C, int64_t L, int64_t Lo, int k) {
    __shared__ float v_s[kTailOut * 8];
    const int64_t nc = blockIdx.x;
    const int c = (int)(nc % C);
    const float s = scale[c], t = shift[c], pr = pre ? pre[c] : 0.0f;
    const int64_t j0 = (int64_t)blockIdx.y * kTailOut;
    const int outs = (int)((Lo - j0) < kTailOut ? (Lo - j0) : kTailOut), ins = outs * k;
    const float *hp = h + nc * L + j0 * k, *rp = res + nc * L + j0 * k;
    for (int i = threadIdx.x; i < ins; i += kBlock) v_s[i] = act_fwd<1>(hp[i], s, t, pr, 0.0f) + rp[i];
    __syncthreads();
    for (int j = threadIdx.x; j < outs; j += kBlock) {
        float best = -INFINITY;
        int code = 0;
        for (int e = 0; e < k; ++e) {
            const float v = v_s[j * k + e];
            if (v > best || v != v) { best = v; code = e; }
        }
        y[nc * Lo + j0 + j] = best;
        sel[nc * Lo + j0 + j] = (uint8_t)code;
    }
}

__global__ __launch_bounds__(kBlock) void tail_pool1d_backward_kernel(const float *__restrict__ gy, const uint8_t *__restrict__ sel,
                                                                      const float *__restrict__ h, const float *__restrict__ scale,
                                                                      const float *__restrict__ pre, float *__restrict__ g_h,
                                                                      float *__restrict__ g_res, int64_t C, int64_t L, int64_t Lo,
                                                                      int k) {
    __shared__ float g_s[kTailOut];
    __shared__ uint8_t c_s[kTailOut];
    const int64_t nc = blockIdx.x;
    const int c = (int)(nc % C);
    const float s = scale[c], pr = pre ? pre[c] : 0.0f;
    const int64_t j0 = (int64_t)blockIdx.y * kTailOut;
    const bool last = j0 + kTailOut >= Lo;
    const int outs = (int)(last ? (Lo - j0) : kTailOut);
    const int ins = last ? (int)(L - j0 * k) : outs * k;          // the last tile also owns the dropped tail (gradient 0)
    for (int j = threadIdx.x; j < outs; j += kBlock) {
        g_s[j] = gy[nc * Lo + j0 + j];
        c_s[j] = sel[nc * Lo + j0 + j];
    }
    __syncthreads();
    const int64_t base = nc * L + j0 * k;
    for (int i = threadIdx.x; i < ins; i += kBlock) {
        const int j = i / k, e = i - j * k;
        const float g = (j < outs && (int)c_s[j] == e) ? g_s[j] : 0.0f;
        g_res[base + i] = g;
        g_h[base + i] = act_bwd<1>(g, h[base + i], s, 0.0f, pr, 0.0f);
    }
}

// ---- RawNet3's feature normalisation (src/models/rawnet3.py:80-85): x = log(|y| + eps);  x = x - mean_t(x) -----------------------
// One workgroup per (n, c) row of L <= kRowCap frames, the row in registers: forward reads y once and writes x once (ATen: abs,
// add, log, mean, sub = 5 passes); backward  g_y = (g - mean_t(g)) / (|y| + eps) * sign(y)  likewise (ATen: 9 kernels).
constexpr int kRowPerThread = 32, kRowCap = kBlock * kRowPerThread;

__device__ __forceinline__ float block_sum(float v, float *red_s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                       // red_s may still be read from a previous call
    if (lane == 0) red_s[wave] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) t += red_s[w];      // same order in every thread
    return t;
}

template <bool BWD>
__global__ __launch_bounds__(kBlock) void log_meannorm_kernel(const float *__restrict__ g, const float *__restrict__ y, float eps,
                                                              float *__restrict__ out, int64_t L) {
    __shared__ float red_s[kBlock / 64];
    const int64_t row = blockIdx.x;
    const float *yp = y + row * L;
    const float *gp = BWD ? g + row * L : nullptr;
    float *op = out + row * L;
    float v[kRowPerThread];
    float part = 0.0f;
#pragma unroll
    for (int k = 0; k < kRowPerThread; ++k) {
        const int64_t i = threadIdx.x + (int64_t)k * kBlock;
        v[k] = 0.0f;
        if (i < L) {
            v[k] = BWD ? gp[i] : logf(fabsf(yp[i]) + eps);
            part += v[k];
        }
    }
    const float mean = block_sum(part, red_s) / (float)L;
#pragma unroll
    for (int k = 0; k < kRowPerThread; ++k) {
        const int64_t i = threadIdx.x + (int64_t)k * kBlock;
        if (i < L) {
            if (BWD) {
                const float yv = yp[i];
                const float sg = yv > 0.0f ? 1.0f : (yv < 0.0f ? -1.0f : (yv == 0.0f ? 0.0f : yv));      // torch.sign; NaN stays NaN
                op[i] = (v[k] - mean) / (fabsf(yv) + eps) * sg;
            } else {
                op[i] = v[k] - mean;
            }
        }
    }
}

// ---- RawNet3's AFMS (src/models/rawnet3.py:161-182): y = sigmoid(fc(mean_t x)); out = (x + alpha[c]) * y[n, c] ----------------------
// Row kernels (one workgroup per (n, c) row of L <= kRowCap frames, the row in registers):
//   MODE 0  r[row]   = mean_t x                                            (the gate's input; fc and sigmoid stay torch ops on (N, C))
//   MODE 1  out      = (x + alpha[c]) * y[row]
//   MODE 2  r[row]   = sum_t g * (x + alpha[c])                            (d out / d y, summed over the row)
//   MODE 3  out      = g * y[row] + m[row]                                 (d out / d x plus the mean's share m = d mean / L)
template <int MODE>
__global__ __launch_bounds__(kBlock) void afms_row_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ alpha, const float *__restrict__ r0,
                                                          const float *__restrict__ r1, float *__restrict__ out, int64_t C, int64_t L) {
    __shared__ float red_s[kBlock / 64];
    const int64_t row = blockIdx.x;
    const float *ap = a + row * L;
    const float *bp = MODE == 2 ? b + row * L : nullptr;
    const float al = (MODE == 1 || MODE == 2) ? alpha[row % C] : 0.0f;
    const float s0 = (MODE == 1 || MODE == 3) ? r0[row] : 0.0f, s1 = MODE == 3 ? r1[row] : 0.0f;
    float part = 0.0f;
    for (int64_t i = threadIdx.x; i < L; i += kBlock) {
        const float v = ap[i];
        if (MODE == 0) part += v;
        if (MODE == 1) out[row * L + i] = (v + al) * s0;
        if (MODE == 2) part += v * (bp[i] + al);
        if (MODE == 3) out[row * L + i] = v * s0 + s1;
    }
    if (MODE == 0 || MODE == 2) {
        const float t = block_sum(part, red_s);
        if (threadIdx.x == 0) out[row] = MODE == 0 ? t / (float)L : t;
    }
}

// ---- RawNet3's attentive statistics (src/models/rawnet3.py:131-132): mu = sum_t x w,  m2 = sum_t x^2 w  per (n, c) row ---------------
// One wave per row (L = 429 frames): forward reads x and w once; backward  g_x = g_mu w + 2 g_m2 x w,  g_w = g_mu x + g_m2 x^2.
constexpr int kRowsPerBlock = kBlock / 64;

__global__ __launch_bounds__(kBlock) void wstats_forward_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                float *__restrict__ mu, float *__restrict__ m2, int64_t rows, int64_t L) {
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float a = 0.0f, b = 0.0f;
    for (int64_t i = lane; i < L; i += 64) {
        const float xv = x[row * L + i], wv = w[row * L + i];
        a += xv * wv;
        b += (xv * xv) * wv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    if (lane == 0) {
        mu[row] = a;
        m2[row] = b;
    }
}

__global__ __launch_bounds__(kBlock) void wstats_backward_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                 const float *__restrict__ gmu, const float *__restrict__ gm2,
                                                                 float *__restrict__ gx, float *__restrict__ gw, int64_t rows, int64_t L) {
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float ga = gmu[row], gb = gm2[row];
    for (int64_t i = lane; i < L; i += 64) {
        const float xv = x[row * L + i], wv = w[row * L + i];
        gx[row * L + i] = ga * wv + (2.0f * gb) * (xv * wv);
        gw[row * L + i] = ga * xv + gb * (xv * xv);
    }
}

inline bool pool_dims_ok(int64_t N, int64_t C, int64_t H, int64_t W) {
    // planes on grid.x, tiles of a plane on grid.y (<= 65535 workgroups of 256 threads per plane)
    return N >= 0 && C >= 0 && H >= 0 && W >= 0 && N * C <= 0x7fffffffLL && H * ((W + 3) / 4) <= 65535LL * kBlock;
}


// ---- SpecRNet's attention gate on (N, C): the Linear + sigmoid forward and its whole backward as one launch each (round 3) ----
// (specrnet.py:145-149: gate = sigmoid(fc(mean_hw x)); around it autograd ran addmm + sigmoid forward and sum, mul, rsub, mul,
// mm, div backward — nine launches of 2-5 us on (128, 20..64) tensors, three times per iteration.)  One workgroup per sample,
// thread c = output channel; C <= 256.
__global__ __launch_bounds__(256) void gate_fc_forward_kernel(const float *__restrict__ mean, const float *__restrict__ w,
                                                              const float *__restrict__ bias, float *__restrict__ gate, int C) {
    __shared__ float m_s[256];
    const int n = blockIdx.x, c = threadIdx.x;
    if (c < C) m_s[c] = mean[(int64_t)n * C + c];
    __syncthreads();
    if (c >= C) return;
    float acc = bias ? bias[c] : 0.0f;
    const float *wr = w + (int64_t)c * C;
    for (int k = 0; k < C; ++k) acc = fmaf(wr[k], m_s[k], acc);
    gate[(int64_t)n * C + c] = 1.0f / (1.0f + expf(-acc));
}

// g_mean[n][k] = inv_hw * sum_c (ggate[n][c] * gate (1 - gate)) * w[c][k],  ggate[n][c] = sum over the partial blocks (fixed order)
__global__ __launch_bounds__(256) void gate_fc_backward_kernel(const float *__restrict__ partial, int blocks,
                                                               const float *__restrict__ gate, const float *__restrict__ w,
                                                               float inv_hw, float *__restrict__ g_mean, int C) {
    __shared__ float t_s[256];
    const int n = blockIdx.x, c = threadIdx.x;
    if (c < C) {
        const float *pr = partial + ((int64_t)n * C + c) * blocks;
        float s = 0.0f;
        for (int b = 0; b < blocks; ++b) s += pr[b];
        const float g = gate[(int64_t)n * C + c];
        t_s[c] = s * g * (1.0f - g);
    }
    __syncthreads();
    if (c >= C) return;
    float acc = 0.0f;
    for (int k = 0; k < C; ++k) acc = fmaf(t_s[k], w[(int64_t)k * C + c], acc);
    g_mean[(int64_t)n * C + c] = acc * inv_hw;
}

}  // namespace

extern "C" {

int advstep_affine_act_forward_f32(const float *x, const float *scale, const float *shift, const float *pre, float *y,
                                   int64_t N, int64_t C, int64_t P, int mode, float slope, advstep_stream_t stream) {
    return launch_affine<false>(nullptr, x, scale, shift, pre, y, N, C, P, mode, slope, as_stream(stream));
}

int advstep_affine_act_backward_f32(const float *gy, const float *x, const float *scale, const float *shift,
                                    const float *pre, float *gx, int64_t N, int64_t C, int64_t P, int mode, float slope,
                                    advstep_stream_t stream) {
    return launch_affine<true>(gy, x, scale, shift, pre, gx, N, C, P, mode, slope, as_stream(stream));
}

int advstep_res2net_link_forward_f32(const float *h, const float *scale, const float *shift, const float *pre, float *y,
                                     int64_t y_bs, const float *other, int64_t other_bs, float *z, int64_t N, int64_t C, int64_t P,
                                     advstep_stream_t stream) {
    if (N < 0 || C < 0 || P < 0 || y_bs < C * P || (other && other_bs < C * P)) return ADVSTEP_EINVAL;
    if (N * C * P == 0) return ADVSTEP_OK;
    if (!h || !scale || !shift || !y || ((other != nullptr) != (z != nullptr))) return ADVSTEP_EINVAL;
    return launch_link<false>(h, other, other_bs, nullptr, 0, scale, shift, pre, y, y_bs, z, N, C, P, as_stream(stream));
}

int advstep_res2net_link_backward_f32(const float *g1, int64_t g1_bs, const float *g2, int64_t g2_bs, const float *h,
                                      const float *scale, const float *shift, const float *pre, float *gx, int64_t N, int64_t C,
                                      int64_t P, advstep_stream_t stream) {
    if (N < 0 || C < 0 || P < 0 || g1_bs < C * P || (g2 && g2_bs < C * P)) return ADVSTEP_EINVAL;
    if (N * C * P == 0) return ADVSTEP_OK;
    if (!g1 || !h || !scale || !shift || !gx) return ADVSTEP_EINVAL;
    return launch_link<true>(h, g1, g1_bs, g2, g2_bs, scale, shift, pre, gx, C * P, nullptr, N, C, P, as_stream(stream));
}

int advstep_add_maxpool2_forward_f32(const float *a, const float *b, const float *bias, float *y, uint8_t *sel, int64_t N,
                                     int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    if (!pool_dims_ok(N, C, H, W)) return ADVSTEP_EINVAL;
    const int64_t Ho = H / 2, Wo = W / 2;
    if (N * C * Ho * Wo == 0) return ADVSTEP_OK;
    if (!a || !y || !sel) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)(N * C), (unsigned)ceil_div(Ho * ((Wo + 1) / 2), kBlock)), block(kBlock);
    hipLaunchKernelGGL(pool2_forward_kernel<0>, grid, block, 0, as_stream(stream), a, b, bias, (const float *)nullptr, y, sel, C,
                       (int)H, (int)W, (float *)nullptr);
    return status_after_launch();
}

int advstep_add_maxpool1d_forward_f32(const float *a, const float *b, float *y, uint8_t *sel, int64_t N, int64_t C, int64_t L,
                                     int64_t k, advstep_stream_t stream) {
    if (N < 0 || C < 0 || L < 0 || k < 2 || k > 8 || N * C > 0x7fffffffLL) return ADVSTEP_EINVAL;
    const int64_t Lo = L / k;
    if (N * C * Lo == 0) return ADVSTEP_OK;
    if (!a || !y || !sel || ceil_div(Lo, kBlock) > 65535) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)(N * C), (unsigned)ceil_div(Lo, kBlock)), block(kBlock);
    hipLaunchKernelGGL(pool1d_forward_kernel, grid, block, 0, as_stream(stream), a, b, y, sel, L, Lo, (int)k);
    return status_after_launch();
}

int advstep_maxpool1d_backward_f32(const float *gy, const uint8_t *sel, float *g, int64_t N, int64_t C, int64_t L, int64_t k,
                                   advstep_stream_t stream) {
    if (N < 0 || C < 0 || L < 0 || k < 2 || k > 8 || N * C > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (N * C * L == 0) return ADVSTEP_OK;
    const int64_t Lo = L / k;
    if (!g || ceil_div(Lo, kBlock) > 65535) return ADVSTEP_EINVAL;
    if (Lo == 0)
        return hipMemsetAsync(g, 0, (size_t)(N * C * L) * sizeof(float), as_stream(stream)) == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH;
    if (!gy || !sel) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)(N * C), (unsigned)ceil_div(Lo, kBlock)), block(kBlock);
    hipLaunchKernelGGL(pool1d_backward_kernel, grid, block, 0, as_stream(stream), gy, sel, g, L, Lo, (int)k);
    return status_after_launch();
}

int advstep_afms_row_f32(int mode, const float *a, const float *b, const float *alpha, const float *r0, const float *r1, float *out,
                         int64_t rows, int64_t C, int64_t L, advstep_stream_t stream) {
    if (mode < 0 || mode > 3 || rows < 0 || C <= 0 || L < 0 || rows > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (rows == 0 || (L == 0 && (mode == 1 || mode == 3))) return ADVSTEP_OK;
    if (!a || !out || (mode == 2 && !b) || ((mode == 1 || mode == 2) && !alpha) || ((mode == 1 || mode == 3) && !r0) || (mode == 3 && !r1))
        return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)rows), block(kBlock);
    hipStream_t st = as_stream(stream);
    if (mode == 0) hipLaunchKernelGGL(afms_row_kernel<0>, grid, block, 0, st, a, b, alpha, r0, r1, out, C, L);
    else if (mode == 1) hipLaunchKernelGGL(afms_row_kernel<1>, grid, block, 0, st, a, b, alpha, r0, r1, out, C, L);
    else if (mode == 2) hipLaunchKernelGGL(afms_row_kernel<2>, grid, block, 0, st, a, b, alpha, r0, r1, out, C, L);
    else hipLaunchKernelGGL(afms_row_kernel<3>, grid, block, 0, st, a, b, alpha, r0, r1, out, C, L);
    return status_after_launch();
}

int advstep_gate_maxpool2_backward_gate_f32(const float *gy, const uint8_t *sel, const float *x, float *ggate_partial, int64_t N,
                                            int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    if (!pool_dims_ok(N, C, H, W)) return ADVSTEP_EINVAL;
    if (N * C * H * W == 0) return ADVSTEP_OK;
    if (!ggate_partial || !x || ((H / 2) * (W / 2) > 0 && (!gy || !sel))) return ADVSTEP_EINVAL;
    const int blocks = (int)advstep_gate_maxpool2_blocks(H, W);
    const dim3 grid((unsigned)(N * C), (unsigned)blocks), block(kBlock);
    hipLaunchKernelGGL(pool2_backward_kernel<true>, grid, block, 0, as_stream(stream), gy, sel, x, x, (float *)nullptr, ggate_partial,
                       (int)H, (int)W, blocks, (const float *)nullptr);
    return status_after_launch();
}

int advstep_gate_maxpool2_backward_input_f32(const float *gy, const uint8_t *sel, const float *gate, const float *addc, float *gx,
                                             int64_t N, int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    if (!pool_dims_ok(N, C, H, W)) return ADVSTEP_EINVAL;
    if (N * C * H * W == 0) return ADVSTEP_OK;
    if (!gx || !gate || ((H / 2) * (W / 2) > 0 && (!gy || !sel))) return ADVSTEP_EINVAL;
    const int blocks = (int)advstep_gate_maxpool2_blocks(H, W);
    const dim3 grid((unsigned)(N * C), (unsigned)blocks), block(kBlock);
    hipLaunchKernelGGL(pool2_backward_kernel<true>, grid, block, 0, as_stream(stream), gy, sel, (const float *)nullptr, gate, gx,
                       (float *)nullptr, (int)H, (int)W, blocks, addc);
    return status_after_launch();
}

int advstep_weighted_stats_forward_f32(const float *x, const float *w, float *mu, float *m2, int64_t rows, int64_t L,
                                       advstep_stream_t stream) {
    if (rows < 0 || L < 0 || rows > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (rows == 0) return ADVSTEP_OK;
    if (!x || !w || !mu || !m2) return ADVSTEP_EINVAL;
    hipLaunchKernelGGL(wstats_forward_kernel, dim3((unsigned)ceil_div(rows, kRowsPerBlock)), dim3(kBlock), 0, as_stream(stream), x, w, mu,
                       m2, rows, L);
    return status_after_launch();
}

int advstep_weighted_stats_backward_f32(const float *x, const float *w, const float *gmu, const float *gm2, float *gx, float *gw,
                                        int64_t rows, int64_t L, advstep_stream_t stream) {
    if (rows < 0 || L < 0 || rows > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (rows * L == 0) return ADVSTEP_OK;
    if (!x || !w || !gmu || !gm2 || !gx || !gw) return ADVSTEP_EINVAL;
    hipLaunchKernelGGL(wstats_backward_kernel, dim3((unsigned)ceil_div(rows, kRowsPerBlock)), dim3(kBlock), 0, as_stream(stream), x, w,
                       gmu, gm2, gx, gw, rows, L);
    return status_after_launch();
}

int advstep_log_meannorm_max_length(void) { return kRowCap; }

int advstep_log_meannorm_forward_f32(const float *y, float eps, float *x, int64_t rows, int64_t L, advstep_stream_t stream) {
    if (rows < 0 || L < 0 || L > kRowCap || rows > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (rows * L == 0) return ADVSTEP_OK;
    if (!y || !x) return ADVSTEP_EINVAL;
    hipLaunchKernelGGL(log_meannorm_kernel<false>, dim3((unsigned)rows), dim3(kBlock), 0, as_stream(stream), (const float *)nullptr, y, eps,
                       x, L);
    return status_after_launch();
}

int advstep_log_meannorm_backward_f32(const float *gx, const float *y, float eps, float *gy, int64_t rows, int64_t L,
                                      advstep_stream_t stream) {
    if (rows < 0 || L < 0 || L > kRowCap || rows > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (rows * L == 0) return ADVSTEP_OK;
    if (!gx || !y || !gy) return ADVSTEP_EINVAL;
    hipLaunchKernelGGL(log_meannorm_kernel<true>, dim3((unsigned)rows), dim3(kBlock), 0, as_stream(stream), gx, y, eps, gy, L);
    return status_after_launch();
}

int advstep_tail_pool1d_forward_f32(const float *h, const float *res, const float *scale, const float *shift, const float *pre,
                                    float *y, uint8_t *sel, int64_t N, int64_t C, int64_t L, int64_t k, advstep_stream_t stream) {
    if (N < 0 || C < 0 || L < 0 || k < 2 || k > 8 || N * C > 0x7fffffffLL) return ADVSTEP_EINVAL;
    const int64_t Lo = L / k;
    if (N * C * Lo == 0) return ADVSTEP_OK;
    if (!h || !res || !scale || !shift || !y || !sel || ceil_div(Lo, kTailOut) > 65535) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)(N * C), (unsigned)ceil_div(Lo, kTailOut)), block(kBlock);
    hipLaunchKernelGGL(tail_pool1d_forward_kernel, grid, block, 0, as_stream(stream), h, res, scale, shift, pre, y, sel, C, L, Lo, (int)k);
    return status_after_launch();
}

int advstep_tail_pool1d_backward_f32(const float *gy, const uint8_t *sel, const float *h, const float *scale, const float *pre,
                                     float *g_h, float *g_res, int64_t N, int64_t C, int64_t L, int64_t k, advstep_stream_t stream) {
    if (N < 0 || C < 0 || L < 0 || k < 2 || k > 8 || N * C > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (N * C * L == 0) return ADVSTEP_OK;
    const int64_t Lo = L / k;
    if (!g_h || !g_res || ceil_div(Lo, kTailOut) > 65535) return ADVSTEP_EINVAL;
    if (Lo == 0) {
        const size_t bytes = (size_t)(N * C * L) * sizeof(float);
        return (hipMemsetAsync(g_h, 0, bytes, as_stream(stream)) == hipSuccess && hipMemsetAsync(g_res, 0, bytes, as_stream(stream)) == hipSuccess)
                   ? ADVSTEP_OK : ADVSTEP_ELAUNCH;
    }
    if (!gy || !sel || !h || !scale) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)(N * C), (unsigned)ceil_div(Lo, kTailOut)), block(kBlock);
    hipLaunchKernelGGL(tail_pool1d_backward_kernel, grid, block, 0, as_stream(stream), gy, sel, h, scale, pre, g_h, g_res, C, L, Lo, (int)k);
    return status_after_launch();
}

int advstep_maxpool2_backward_f32(const float *gy, const uint8_t *sel, float *g, int64_t N, int64_t C, int64_t H, int64_t W,
                                  advstep_stream_t stream) {
    if (!pool_dims_ok(N, C, H, W)) return ADVSTEP_EINVAL;
    if (N * C * H * W == 0) return ADVSTEP_OK;
    if (!g || ((H / 2) * (W / 2) > 0 && (!gy || !sel))) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)(N * C), (unsigned)ceil_div(H * ((W + 3) / 4), kBlock)), block(kBlock);
    hipLaunchKernelGGL(pool2_backward_kernel<false>, grid, block, 0, as_stream(stream), gy, sel, (const float *)nullptr,
                       (const float *)nullptr, g, (float *)nullptr, (int)H, (int)W, 0, (const float *)nullptr);
    return status_after_launch();
}

size_t advstep_gate_maxpool2_blocks(int64_t H, int64_t W) {
    if (H <= 0 || W <= 0) return 0;
    return (size_t)ceil_div(H * ((W + 3) / 4), kBlock);
}

int advstep_gate_maxpool2_forward_xw_f32(const float *x, const float *gate, float *y, uint8_t *sel, float *xw, int64_t N,
                                         int64_t C, int64_t H, int64_t W, advstep_stream_t stream) {
    if (!pool_dims_ok(N, C, H, W)) return ADVSTEP_EINVAL;
    const int64_t Ho = H / 2, Wo = W / 2;
    if (N * C * Ho * Wo == 0) return ADVSTEP_OK;
    if (!x || !gate || !y || !sel) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)(N * C), (unsigned)ceil_div(Ho * ((Wo + 1) / 2), kBlock)), block(kBlock);
    hipLaunchKernelGGL(pool2_forward_kernel<1>, grid, block, 0, as_stream(stream), x, (const float *)nullptr,
                       (const float *)nullptr, gate, y, sel, C, (int)H, (int)W, xw);
    return status_after_launch();
}

int advstep_gate_maxpool2_forward_f32(const float *x, const float *gate, float *y, uint8_t *sel, int64_t N, int64_t C,
                                      int64_t H, int64_t W, advstep_stream_t stream) {
    return advstep_gate_maxpool2_forward_xw_f32(x, gate, y, sel, nullptr, N, C, H, W, stream);
}

int advstep_gate_maxpool2_backward_gate_pooled_f32(const float *gy, const float *xw, float *ggate, int64_t N, int64_t C, int64_t H,
                                                   int64_t W, advstep_stream_t stream) {
    if (!pool_dims_ok(N, C, H, W)) return ADVSTEP_EINVAL;
    if (N * C == 0) return ADVSTEP_OK;
    const int64_t L = (H / 2) * (W / 2);
    if (!ggate || (L > 0 && (!gy || !xw))) return ADVSTEP_EINVAL;
    hipLaunchKernelGGL(gate_grad_pooled_kernel, dim3((unsigned)(N * C)), dim3(kBlock), 0, as_stream(stream), gy, xw, ggate, L);
    return status_after_launch();
}

int advstep_gate_maxpool2_backward_f32(const float *gy, const uint8_t *sel, const float *x, const float *gate, float *gx,
                                       float *ggate_partial, int64_t N, int64_t C, int64_t H, int64_t W,
                                       advstep_stream_t stream) {
    if (!pool_dims_ok(N, C, H, W)) return ADVSTEP_EINVAL;
    if (N * C * H * W == 0) return ADVSTEP_OK;
    if (!gx || !ggate_partial || !x || !gate || ((H / 2) * (W / 2) > 0 && (!gy || !sel))) return ADVSTEP_EINVAL;
    const int blocks = (int)advstep_gate_maxpool2_blocks(H, W);
    const dim3 grid((unsigned)(N * C), (unsigned)blocks), block(kBlock);
    hipLaunchKernelGGL(pool2_backward_kernel<true>, grid, block, 0, as_stream(stream), gy, sel, x, gate, gx, ggate_partial, (int)H,
                       (int)W, blocks, (const float *)nullptr);
    return status_after_launch();
}

int advstep_gate_fc_forward_f32(const float *mean, const float *w, const float *bias, float *gate, int64_t N, int64_t C,
                                advstep_stream_t stream) {
    if (N < 0 || C < 1 || C > 256 || N > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (N == 0) return ADVSTEP_OK;
    if (!mean || !w || !gate) return ADVSTEP_EINVAL;
    hipLaunchKernelGGL(gate_fc_forward_kernel, dim3((unsigned)N), dim3(256), 0, as_stream(stream), mean, w, bias, gate, (int)C);
    return status_after_launch();
}

int advstep_gate_fc_backward_f32(const float *ggate_partial, int64_t blocks, const float *gate, const float *w, float inv_hw,
                                 float *g_mean, int64_t N, int64_t C, advstep_stream_t stream) {
    if (N < 0 || C < 1 || C > 256 || blocks < 1 || blocks > 0x7fffffffLL || N > 0x7fffffffLL) return ADVSTEP_EINVAL;
    if (N == 0) return ADVSTEP_OK;
    if (!ggate_partial || !gate || !w || !g_mean) return ADVSTEP_EINVAL;
    hipLaunchKernelGGL(gate_fc_backward_kernel, dim3((unsigned)N), dim3(256), 0, as_stream(stream), ggate_partial, (int)blocks, gate, w,
                       inv_hw, g_mean, (int)C);
    return status_after_launch();
}

}  // extern "C"
