// advstep.hip — gfx950 (MI355X / CDNA4) kernels + the C ABI declared in include/advstep.h.
//
// Every kernel here is HBM-bound streaming or row-reduction work over (B, T) float32 waveforms:
//   * 16 B per lane (float4) coalesced loads, 4 independent float4 per input stream per thread in flight
//     (a wave64 instruction moves 1 KiB; a 256-thread workgroup owns one 16 KiB tile per stream);
//   * row reductions: wave64 xor-shuffles -> 4-entry LDS -> one partial per (row, tile) in the caller's
//     workspace -> re-reduced by every workgroup of the row in the consuming pass (fixed order, no float
//     atomics: run-to-run deterministic);
//   * the (row, tile) -> workgroup id map is identical in the producing and consuming pass of a row
//     operation, so with the observed id % 8 -> XCD placement a tile is re-read on the XCD whose L2 saw it;
//   * arithmetic follows the reference expression by expression (see include/advstep.h): this file is built
//     with -ffp-contract=off, divisions are IEEE (hipcc default for f32), clamps propagate NaN.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared (see build.py).

#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "advstep.h"

namespace {

constexpr int kBlock = 256;                        // 4 wave64 per workgroup
constexpr int kVecs = 4;                           // float4 per thread per stream
constexpr int kTileVec = kBlock * kVecs;           // 1024 float4 per workgroup tile
constexpr int kTile = kTileVec * 4;                // 4096 floats (16 KiB) per stream per tile
constexpr int kMaxGrid = 256 * 16;                 // flat kernels: grid-stride beyond 16 tiles per CU

// ---------------------------------------------------------------------------------------------------------
// scalar semantics shared by all kernels
// ---------------------------------------------------------------------------------------------------------

// torch.sign: (0 < g) - (g < 0); NaN and +-0 give 0.
__device__ __forceinline__ float sgn(float g) { return (float)(0.0f < g) - (float)(g < 0.0f); }

// torch.clamp(v, lo, hi) = min(max(v, lo), hi), NaN in v propagates.
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    v = (v < lo) ? lo : v;
    return (v > hi) ? hi : v;
}

// torch.min(a, b) for tensors: NaN propagates.
__device__ __forceinline__ float min_nan(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    return a < b ? a : b;
}
__device__ __forceinline__ float max_nan(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    return a > b ? a : b;
}

__device__ __forceinline__ float fgsm_elem(float x, float g, float eps, float lo, float hi) {
    return clampf(x + eps * sgn(g), lo, hi);
}

__device__ __forceinline__ float pgd_linf_elem(float a, float g, float x, float alpha, float eps, float lo,
                                               float hi) {
    a = a + alpha * sgn(g);
    float d = clampf(a - x, -eps, eps);
    return clampf(x + d, lo, hi);
}

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter-based: the same stream on the CPU oracle and on the device.
// ---------------------------------------------------------------------------------------------------------

struct Quad {
    uint32_t v[4];
};

__device__ __forceinline__ Quad philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Quad{{c0, c1, c2, c3}};
}

__device__ __forceinline__ float u01(uint32_t bits) { return (float)(bits >> 8) * 5.9604644775390625e-08f; }
__device__ __forceinline__ float u01_open0(uint32_t bits) {  // (0, 1]
    return (float)((bits >> 8) + 1u) * 5.9604644775390625e-08f;
}

// 4 uniforms in [-eps, eps): u * (eps - (-eps)) + (-eps)
__device__ __forceinline__ float4 philox_uniform4(uint64_t q, uint64_t seed, uint64_t offset, float eps) {
    const Quad r = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32),
                                 (uint32_t)seed, (uint32_t)(seed >> 32));
    const float from = -eps, range = eps - from;
    return make_float4(u01(r.v[0]) * range + from, u01(r.v[1]) * range + from, u01(r.v[2]) * range + from,
                       u01(r.v[3]) * range + from);
}

// 4 standard normals for quad q of row b (Box-Muller on two uniform pairs).
__device__ __forceinline__ float4 philox_normal4(uint32_t q, uint32_t b, uint64_t seed, uint64_t offset) {
    const Quad r = philox4x32_10(q, b, (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
                                 (uint32_t)(seed >> 32));
    // Round 4: the hardware transcendentals (v_log_f32, v_sin_f32 / v_cos_f32: ~1 ulp / ~1e-6 absolute) instead of OCML's
    // correctly-rounded-ish logf / sinf / cosf, which were what this start kernel spent its time in (24 us for 8 B / sample = 0.34
    // of the HBM roofline).  The random start has no reference bit pattern to match (pgdl2.py:55-62 draws from torch's own
    // generator); the oracle twin (oracle/kernels.py, libm) agrees to ~1e-9 after the eps / ||n|| scaling, inside the 1e-7 bound
    // oracle/checked_ops.py applies, and every path of the library (single-pass, repair, two-kernel) calls THIS function.
    const float r0 = sqrtf(-2.0f * __logf(u01_open0(r.v[0])));
    const float r1 = sqrtf(-2.0f * __logf(u01_open0(r.v[2])));
    const float t0 = 6.283185307179586f * u01(r.v[1]);
    const float t1 = 6.283185307179586f * u01(r.v[3]);
    return make_float4(r0 * __cosf(t0), r0 * __sinf(t0), r1 * __cosf(t1), r1 * __sinf(t1));
}

__device__ __forceinline__ float f4_get(const float4 &v, int k) {
    return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
}

// ---------------------------------------------------------------------------------------------------------
// workgroup reductions (wave64 shuffles -> LDS)
// ---------------------------------------------------------------------------------------------------------

struct SumOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct MinOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return min_nan(a, b); }
};
struct MaxOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return max_nan(a, b); }
};

template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, 64));
    return v;
}

// All threads of the workgroup receive the result. `lds` holds >= 4 floats; two calls in a row must use
// different slots (or be separated by the caller) — each call ends with a barrier on its own slot.
template <class Op>
__device__ __forceinline__ float block_reduce(float v, Op op, float *lds) {
    v = wave_reduce(v, op);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    float r = lds[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) r = op(r, lds[w]);
    return r;
}

// Re-reduce the C per-tile partials of one row (every workgroup of the row does this; C is 16 at T = 64 600).
template <class Op>
__device__ __forceinline__ float reduce_partials(const float *__restrict__ part, int C, float identity, Op op,
                                                 float *lds) {
    float v = identity;
    for (int i = threadIdx.x; i < C; i += kBlock) v = op(v, part[i]);
    return block_reduce(v, op, lds);
}

// ---------------------------------------------------------------------------------------------------------
// flat elementwise kernels: n = B*T samples, tile-strided, float4 when all pointers are 16-byte aligned
// ---------------------------------------------------------------------------------------------------------

// Generic driver: NIN input streams, one output stream, Op applied per sample; VECS float4 per thread per stream.
template <int NIN, int VECS, class Op>
__global__ __launch_bounds__(kBlock) void flat_vec_kernel(const float4 *__restrict__ in0,
                                                          const float4 *__restrict__ in1,
                                                          const float4 *__restrict__ in2, float4 *out, int64_t n4,
                                                          int64_t ntiles, Op op) {
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * (kBlock * VECS) + threadIdx.x;
        float4 a[VECS] = {}, b[VECS] = {}, c[VECS] = {};
#pragma unroll
        for (int j = 0; j < VECS; ++j) {
            const int64_t i = base + (int64_t)j * kBlock;
            if (i < n4) {
                a[j] = in0[i];
                if (NIN > 1) b[j] = in1[i];
                if (NIN > 2) c[j] = in2[i];
            }
        }
#pragma unroll
        for (int j = 0; j < VECS; ++j) {
            const int64_t i = base + (int64_t)j * kBlock;
            if (i < n4) {
                float4 o;
                o.x = op(a[j].x, b[j].x, c[j].x);
                o.y = op(a[j].y, b[j].y, c[j].y);
                o.z = op(a[j].z, b[j].z, c[j].z);
                o.w = op(a[j].w, b[j].w, c[j].w);
                out[i] = o;
            }
        }
    }
}

// Scalar twin for the (< 4)-sample tail and for unaligned buffers: samples [begin, n).
template <int NIN, class Op>
__global__ __launch_bounds__(kBlock) void flat_scalar_kernel(const float *__restrict__ in0,
                                                             const float *__restrict__ in1,
                                                             const float *__restrict__ in2, float *out,
                                                             int64_t begin, int64_t n, Op op) {
    for (int64_t i = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float a = in0[i];
        const float b = NIN > 1 ? in1[i] : 0.0f;
        const float c = NIN > 2 ? in2[i] : 0.0f;
        out[i] = op(a, b, c);
    }
}

struct FgsmOp {
    float eps, lo, hi;
    __device__ __forceinline__ float operator()(float x, float g, float) const {
        return fgsm_elem(x, g, eps, lo, hi);
    }
};
struct PgdLinfOp {
    float alpha, eps, lo, hi;
    __device__ __forceinline__ float operator()(float a, float g, float x) const {
        return pgd_linf_elem(a, g, x, alpha, eps, lo, hi);
    }
};
struct AddClampOp {
    float lo, hi;
    __device__ __forceinline__ float operator()(float x, float nz, float) const { return clampf(x + nz, lo, hi); }
};
struct CwInitOp {
    // cw.py:117-122: atanh(x*2 - 1) spelled 0.5*log((1+y)/(1-y))
    __device__ __forceinline__ float operator()(float x, float, float) const {
        const float y = x * 2.0f - 1.0f;
        return 0.5f * logf((1.0f + y) / (1.0f - y));
    }
};

// PGD L-inf random start with in-kernel Philox: quad index == float4 index.
__global__ __launch_bounds__(kBlock) void pgd_linf_init_philox_vec_kernel(const float4 *__restrict__ x, float4 *out,
                                                                          int64_t n4, int64_t ntiles, float eps,
                                                                          float lo, float hi, uint64_t seed,
                                                                          uint64_t offset) {
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileVec + threadIdx.x;
        float4 a[kVecs];
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            const int64_t i = base + (int64_t)j * kBlock;
            if (i < n4) a[j] = x[i];
        }
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            const int64_t i = base + (int64_t)j * kBlock;
            if (i < n4) {
                const float4 nz = philox_uniform4((uint64_t)i, seed, offset, eps);
                float4 o;
                o.x = clampf(a[j].x + nz.x, lo, hi);
                o.y = clampf(a[j].y + nz.y, lo, hi);
                o.z = clampf(a[j].z + nz.z, lo, hi);
                o.w = clampf(a[j].w + nz.w, lo, hi);
                out[i] = o;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void pgd_linf_init_philox_scalar_kernel(const float *__restrict__ x, float *out,
                                                                             int64_t begin, int64_t n, float eps,
                                                                             float lo, float hi, uint64_t seed,
                                                                             uint64_t offset) {
    for (int64_t i = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float4 nz = philox_uniform4((uint64_t)(i >> 2), seed, offset, eps);
        out[i] = clampf(x[i] + f4_get(nz, (int)(i & 3)), lo, hi);
    }
}

// CW Adam step on w (tanh recomputed; see header).
struct AdamScalars {
    float w1;        // 1 - beta1           (lerp weight)
    float beta2;     // beta2
    float omb2;      // 1 - beta2
    float neg_step;  // -(lr / (1 - beta1^t))
    float bc2_sqrt;  // sqrt(1 - beta2^t)
    float eps;
};

__device__ __forceinline__ void cw_adam_elem(float &w, float &m, float &v, float x, float gm, const AdamScalars &s) {
    const float y = tanhf(w);
    const float a = 0.5f * (y + 1.0f);
    const float g = ((2.0f * (a - x) + gm) * 0.5f) * (1.0f - y * y);
    m = m + s.w1 * (g - m);
    v = v * s.beta2 + (s.omb2 * g) * g;
    const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
    w = w + s.neg_step * (m / denom);
}

__global__ __launch_bounds__(kBlock) void cw_adam_vec_kernel(float4 *w, float4 *m, float4 *v,
                                                             const float4 *__restrict__ x,
                                                             const float4 *__restrict__ gm, int64_t n4,
                                                             int64_t ntiles, AdamScalars s) {
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileVec + threadIdx.x;
#pragma unroll 2
        for (int j = 0; j < kVecs; ++j) {
            const int64_t i = base + (int64_t)j * kBlock;
            if (i < n4) {
                float4 W = w[i], M = m[i], V = v[i];
                const float4 X = x[i], G = gm[i];
                cw_adam_elem(W.x, M.x, V.x, X.x, G.x, s);
                cw_adam_elem(W.y, M.y, V.y, X.y, G.y, s);
                cw_adam_elem(W.z, M.z, V.z, X.z, G.z, s);
                cw_adam_elem(W.w, M.w, V.w, X.w, G.w, s);
                w[i] = W;
                m[i] = M;
                v[i] = V;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void cw_adam_scalar_kernel(float *w, float *m, float *v,
                                                                const float *__restrict__ x,
                                                                const float *__restrict__ gm, int64_t begin,
                                                                int64_t n, AdamScalars s) {
    for (int64_t i = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float W = w[i], M = m[i], V = v[i];
        cw_adam_elem(W, M, V, x[i], gm[i], s);
        w[i] = W;
        m[i] = M;
        v[i] = V;
    }
}

// ---------------------------------------------------------------------------------------------------------
// row kernels: grid = (C tiles per row, B rows); tile c of row b covers samples [c*kTile, min(T, (c+1)*kTile))
// VEC = rows are float4-addressable (T % 4 == 0 and 16-byte aligned bases).
// ---------------------------------------------------------------------------------------------------------

// Loads the workgroup's tile of one row into registers (out-of-range lanes get `fill`).
template <bool VEC>
__device__ __forceinline__ void load_tile(const float *__restrict__ row, int64_t T, int tile, float fill,
                                          float4 (&r)[kVecs]) {
    if (VEC) {
        const float4 *row4 = reinterpret_cast<const float4 *>(row);
        const int64_t T4 = T >> 2;
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            const int64_t i = (int64_t)tile * kTileVec + j * kBlock + threadIdx.x;
            r[j] = (i < T4) ? row4[i] : make_float4(fill, fill, fill, fill);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            const int64_t i = ((int64_t)tile * kTileVec + j * kBlock + threadIdx.x) * 4;
            r[j].x = (i + 0 < T) ? row[i + 0] : fill;
            r[j].y = (i + 1 < T) ? row[i + 1] : fill;
            r[j].z = (i + 2 < T) ? row[i + 2] : fill;
            r[j].w = (i + 3 < T) ? row[i + 3] : fill;
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void store_tile(float *row, int64_t T, int tile, const float4 (&r)[kVecs]) {
    if (VEC) {
        float4 *row4 = reinterpret_cast<float4 *>(row);
        const int64_t T4 = T >> 2;
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            const int64_t i = (int64_t)tile * kTileVec + j * kBlock + threadIdx.x;
            if (i < T4) row4[i] = r[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            const int64_t i = ((int64_t)tile * kTileVec + j * kBlock + threadIdx.x) * 4;
            if (i + 0 < T) row[i + 0] = r[j].x;
            if (i + 1 < T) row[i + 1] = r[j].y;
            if (i + 2 < T) row[i + 2] = r[j].z;
            if (i + 3 < T) row[i + 3] = r[j].w;
        }
    }
}

// Is sample k (0..3) of vector j of this thread inside the row?  (only needed where `fill` cannot be neutral)
__device__ __forceinline__ bool in_row(int64_t T, int tile, int j, int k) {
    return ((int64_t)tile * kTileVec + j * kBlock + threadIdx.x) * 4 + k < T;
}

#define ADV_FOR_EACH_LANE(r, expr)      \
    _Pragma("unroll") for (int j = 0; j < kVecs; ++j) { \
        { float &e = r[j].x; const int k = 0; (void)k; expr; } \
        { float &e = r[j].y; const int k = 1; (void)k; expr; } \
        { float &e = r[j].z; const int k = 2; (void)k; expr; } \
        { float &e = r[j].w; const int k = 3; (void)k; expr; } \
    }

// ---- a1: to_minmax ----------------------------------------------------------------------------------------

template <bool VEC>
__global__ __launch_bounds__(kBlock) void minmax_partial_kernel(const float *__restrict__ x, int64_t T,
                                                                float *__restrict__ pmin, float *__restrict__ pmax) {
    __shared__ float lds[8];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    const float *row = x + b * T;
    // neutral fill: the row's first sample (always in range) keeps NaN propagation intact
    const float fill = row[0];
    float4 r[kVecs];
    load_tile<VEC>(row, T, tile, fill, r);
    float lo = fill, hi = fill;
    ADV_FOR_EACH_LANE(r, lo = min_nan(lo, e); hi = max_nan(hi, e));
    lo = block_reduce(lo, MinOp(), lds);
    hi = block_reduce(hi, MaxOp(), lds + 4);
    if (threadIdx.x == 0) {
        pmin[b * C + tile] = lo;
        pmax[b * C + tile] = hi;
    }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void minmax_apply_kernel(const float *__restrict__ x, float *x01,
                                                              float *__restrict__ mn_out, float *__restrict__ mx_out,
                                                              int64_t T, const float *__restrict__ pmin,
                                                              const float *__restrict__ pmax) {
    __shared__ float lds[8];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 r[kVecs];
    load_tile<VEC>(x + b * T, T, tile, 0.0f, r);
    const float first = pmin[b * C];
    const float mn = reduce_partials(pmin + b * C, C, first, MinOp(), lds);
    const float mx = reduce_partials(pmax + b * C, C, pmax[b * C], MaxOp(), lds + 4);
    const float range = mx - mn;
    ADV_FOR_EACH_LANE(r, e = (e - mn) / range);
    store_tile<VEC>(x01 + b * T, T, tile, r);
    if (tile == 0 && threadIdx.x == 0) {
        mn_out[b] = mn;
        mx_out[b] = mx;
    }
}

// ---- a2: revert_minmax ------------------------------------------------------------------------------------

template <bool VEC>
__global__ __launch_bounds__(kBlock) void minmax_revert_kernel(const float *__restrict__ x01,
                                                               const float *__restrict__ mn,
                                                               const float *__restrict__ mx, float *out, int64_t T) {
    const int tile = blockIdx.x;
    const int64_t b = blockIdx.y;
    float4 r[kVecs];
    load_tile<VEC>(x01 + b * T, T, tile, 0.0f, r);
    const float lo = mn[b];
    const float range = mx[b] - lo;
    ADV_FOR_EACH_LANE(r, e = (e * range) + lo);
    store_tile<VEC>(out + b * T, T, tile, r);
}

// ---- row sum of squares (||grad||^2, ||normal||^2) ------------------------------------------------------------

template <bool VEC>
__global__ __launch_bounds__(kBlock) void sumsq_partial_kernel(const float *__restrict__ g, int64_t T,
                                                               float *__restrict__ part) {
    __shared__ float lds[4];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 r[kVecs];
    load_tile<VEC>(g + b * T, T, tile, 0.0f, r);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kVecs; ++j) s += (r[j].x * r[j].x + r[j].y * r[j].y) + (r[j].z * r[j].z + r[j].w * r[j].w);
    s = block_reduce(s, SumOp(), lds);
    if (threadIdx.x == 0) part[b * C + tile] = s;
}

// ---- a6: PGD-L2 step ------------------------------------------------------------------------------------------

// pass 2: gn from the grad partials; a = adv + alpha * (g / gn); d = a - orig; partial sum d^2
template <bool VEC>
__global__ __launch_bounds__(kBlock) void pgd_l2_delta_kernel(const float *__restrict__ adv,
                                                              const float *__restrict__ grad,
                                                              const float *__restrict__ orig, int64_t T, float alpha,
                                                              float eps_div, const float *__restrict__ gpart,
                                                              float *__restrict__ dpart, float *__restrict__ gnorm) {
    __shared__ float lds[8];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 a[kVecs], g[kVecs], x[kVecs];
    load_tile<VEC>(adv + b * T, T, tile, 0.0f, a);
    load_tile<VEC>(grad + b * T, T, tile, 0.0f, g);
    load_tile<VEC>(orig + b * T, T, tile, 0.0f, x);
    const float gsq = reduce_partials(gpart + b * C, C, 0.0f, SumOp(), lds);
    const float gn_raw = sqrtf(gsq);
    const float gn = gn_raw + eps_div;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        float4 d;
        d.x = (a[j].x + alpha * (g[j].x / gn)) - x[j].x;
        d.y = (a[j].y + alpha * (g[j].y / gn)) - x[j].y;
        d.z = (a[j].z + alpha * (g[j].z / gn)) - x[j].z;
        d.w = (a[j].w + alpha * (g[j].w / gn)) - x[j].w;
        // out-of-row lanes: a = g = x = 0 -> d = 0 when gn != 0; mask explicitly so gn == 0 / NaN cannot leak
        if (!in_row(T, tile, j, 0)) d.x = 0.0f;
        if (!in_row(T, tile, j, 1)) d.y = 0.0f;
        if (!in_row(T, tile, j, 2)) d.z = 0.0f;
        if (!in_row(T, tile, j, 3)) d.w = 0.0f;
        s += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    s = block_reduce(s, SumOp(), lds + 4);
    if (threadIdx.x == 0) {
        dpart[b * C + tile] = s;
        if (tile == 0 && gnorm) gnorm[b] = gn_raw;
    }
}

// pass 3: recompute d, f = min((1/dn) * eps, 1), out = clamp(orig + d * f, lo, hi)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void pgd_l2_project_kernel(const float *__restrict__ adv,
                                                                const float *__restrict__ grad,
                                                                const float *__restrict__ orig, float *out, int64_t T,
                                                                float alpha, float eps, float eps_div, float lo,
                                                                float hi, const float *__restrict__ gpart,
                                                                const float *__restrict__ dpart,
                                                                float *__restrict__ dnorm) {
    __shared__ float lds[8];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 a[kVecs], g[kVecs], x[kVecs];
    load_tile<VEC>(adv + b * T, T, tile, 0.0f, a);
    load_tile<VEC>(grad + b * T, T, tile, 0.0f, g);
    load_tile<VEC>(orig + b * T, T, tile, 0.0f, x);
    const float gn = sqrtf(reduce_partials(gpart + b * C, C, 0.0f, SumOp(), lds)) + eps_div;
    const float dn = sqrtf(reduce_partials(dpart + b * C, C, 0.0f, SumOp(), lds + 4));
    const float f = min_nan((1.0f / dn) * eps, 1.0f);
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        float4 d;
        d.x = (a[j].x + alpha * (g[j].x / gn)) - x[j].x;
        d.y = (a[j].y + alpha * (g[j].y / gn)) - x[j].y;
        d.z = (a[j].z + alpha * (g[j].z / gn)) - x[j].z;
        d.w = (a[j].w + alpha * (g[j].w / gn)) - x[j].w;
        a[j].x = clampf(x[j].x + d.x * f, lo, hi);
        a[j].y = clampf(x[j].y + d.y * f, lo, hi);
        a[j].z = clampf(x[j].z + d.z * f, lo, hi);
        a[j].w = clampf(x[j].w + d.w * f, lo, hi);
    }
    store_tile<VEC>(out + b * T, T, tile, a);
    if (tile == 0 && threadIdx.x == 0 && dnorm) dnorm[b] = dn;
}


// ---- a6 in ONE pass: the row's workgroups keep their tiles in registers across the two row reductions ----------------------
// 16 B per sample (adv, grad, orig read once; out written once) instead of 32.  A row is 16 workgroups (T = 64 600); its
// two norms are exchanged INSIDE the launch through 8-byte {tag, value} granules (one per workgroup and phase, written by ONE
// agent-scope store, re-read with agent-scope loads until every tag of the row shows the phase: the data is the flag, no
// fence — /opt/skills/guides/cdna_hip_programming.md Guideline 16, form R2).  tag = {call epoch, phase}: the workspace's first
// word counts the single-pass calls that used it (the repair kernel behind every call advances it), so a granule of an earlier
// call — or a word some other user of the scratch left there — never carries this call's tag and nothing has to be cleaned
// between calls (round 5; rounds 2-3 zeroed the granules with a memset node, round 4 in the repair kernel).  The partial sums
// and their re-reduction are the three-kernel path's own (same block_reduce, same order): results are bit-identical to it.
//
// Co-residency.  A spinning workgroup must not wait for one that has no slot.  The host takes this path only when the
// launch fits the device's resident capacity for THIS kernel (CU count x hipOccupancyMaxActiveBlocksPerMultiprocessor,
// queried once per device: l2_resident_capacity) — but capacity is not a guarantee: another stream or process may hold
// slots (two ranks sharing a device, a partitioned device).  So the wait is bounded, and a sweep that does not complete
// does NOT produce a wrong row: the workgroup raises the row's flag in the workspace, poisons its later granules (its
// siblings stop waiting at once) and leaves; a repair kernel queued behind every single-pass launch recomputes flagged rows
// from global memory with the three-kernel path's arithmetic (one workgroup per row, tiles in sequence: same partial sums,
// same re-reduction, same bits) and returns immediately for all other rows.
typedef unsigned long long __attribute__((address_space(1))) gu64;
// Tags of one call: (epoch + 1) << 2 | {1, 2: the two row sums; 3: poison}.  30 bits of epoch; a zero-filled workspace holds no
// valid tag (the smallest is 5), and the flag values below (high bit set) are no tag of the first 2^29 calls either.
constexpr unsigned kPoisonPhase = 3u;
constexpr unsigned kRowFlagged = 0x80000001u;
__device__ __forceinline__ unsigned exchange_tag_base(const unsigned *epoch) {
    return (*epoch + 1u) << 2;
}

// Returns the row sum; *failed (workgroup-uniform, valid after the call) tells that the sweep was abandoned.
__device__ __forceinline__ float row_exchange_sum(unsigned long long *granules, int C, int tile, float mine, unsigned base,
                                                  unsigned phase_no, unsigned spin_limit, float *lds, int *lds_failed) {
    const unsigned phase = base | phase_no, poison_tag = base | kPoisonPhase;
    if (threadIdx.x == 0) {
        *lds_failed = 0;
        __hip_atomic_store((gu64 *)(granules + tile), ((unsigned long long)phase << 32) | __float_as_uint(mine),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float v = 0.0f;
    if (threadIdx.x < 64) {                       // wave 0 sweeps the row's granules: lane i < C reads granule i
        bool done = false;
        for (unsigned spins = 0; spins < spin_limit; ++spins) {
            bool ok = true, poisoned = false;
            unsigned long long x = 0;
            if ((int)threadIdx.x < C) {
                x = __hip_atomic_load((gu64 *)(granules + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(x >> 32) == phase;
                poisoned = (unsigned)(x >> 32) == poison_tag;
            }
            if (__any(poisoned)) break;
            if (__all(ok)) {
                v = (int)threadIdx.x < C ? __uint_as_float((unsigned)x) : 0.0f;
                done = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!done && threadIdx.x == 0) *lds_failed = 1;
    }
    // same reduction as reduce_partials(): thread i holds partial i (C <= 64 on this path), the others the identity
    // (block_reduce's barrier also publishes *lds_failed)
    return block_reduce(v, SumOp(), lds);
}

// A workgroup that gives up: flag the row, poison this workgroup's granules of the phases it will not reach.
__device__ __forceinline__ void row_exchange_abandon(unsigned *fail, int64_t b, unsigned long long *g0, unsigned long long *g1,
                                                     int tile, unsigned base) {
    if (threadIdx.x == 0) {
        __hip_atomic_store((unsigned __attribute__((address_space(1))) *)(fail + b), kRowFlagged, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long poison = (unsigned long long)(base | kPoisonPhase) << 32;
        if (g0) __hip_atomic_store((gu64 *)(g0 + tile), poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g1) __hip_atomic_store((gu64 *)(g1 + tile), poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Register budget: all 2 048 workgroups (8 192 waves) must be resident, i.e. 64 VGPRs per lane, and adv + grad + orig of a tile
// are 48 of them — the three arrays of the batch are 100 MB, three quarters of the chip's register files.  `orig` therefore sits
// in LDS (16 KB per workgroup, 8 workgroups per CU = 128 of the 160 KB) and is read back where it is used; adv and grad stay
// in registers.  (With all three in registers hipcc spilled 15 dwords per lane: 31 MB of scratch writes per launch.)
template <bool VEC>
__global__ __launch_bounds__(kBlock, 8) void pgd_l2_fused_kernel(const float *__restrict__ adv, const float *__restrict__ grad,
                                                                 const float *__restrict__ orig, float *out, int64_t T,
                                                                 float alpha, float eps, float eps_div, float lo, float hi,
                                                                 unsigned long long *__restrict__ gran_g,
                                                                 unsigned long long *__restrict__ gran_d,
                                                                 unsigned *__restrict__ fail,
                                                                 const unsigned *__restrict__ epoch, unsigned spin_limit,
                                                                 float *__restrict__ gnorm, float *__restrict__ dnorm) {
    __shared__ float4 xs[kTileVec];
    __shared__ float lds[12];
    __shared__ int failed;
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 a[kVecs], g[kVecs];
    {
        float4 x[kVecs];
        load_tile<VEC>(orig + b * T, T, tile, 0.0f, x);
#pragma unroll
        for (int j = 0; j < kVecs; ++j) xs[j * kBlock + threadIdx.x] = x[j];      // each thread reads back only its own slots
    }
    load_tile<VEC>(adv + b * T, T, tile, 0.0f, a);
    load_tile<VEC>(grad + b * T, T, tile, 0.0f, g);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kVecs; ++j) s += (g[j].x * g[j].x + g[j].y * g[j].y) + (g[j].z * g[j].z + g[j].w * g[j].w);
    s = block_reduce(s, SumOp(), lds);
    const unsigned base = exchange_tag_base(epoch);
    const float gsq = row_exchange_sum(gran_g + b * C, C, tile, s, base, 1u, spin_limit, lds + 4, &failed);
    if (failed) {
        row_exchange_abandon(fail, b, gran_g + b * C, gran_d + b * C, tile, base);
        return;
    }
    const float gn_raw = sqrtf(gsq);
    const float gn = gn_raw + eps_div;
    s = 0.0f;
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        const float4 x = xs[j * kBlock + threadIdx.x];
        float4 m;
        m.x = (a[j].x + alpha * (g[j].x / gn)) - x.x;
        m.y = (a[j].y + alpha * (g[j].y / gn)) - x.y;
        m.z = (a[j].z + alpha * (g[j].z / gn)) - x.z;
        m.w = (a[j].w + alpha * (g[j].w / gn)) - x.w;
        if (!in_row(T, tile, j, 0)) m.x = 0.0f;
        if (!in_row(T, tile, j, 1)) m.y = 0.0f;
        if (!in_row(T, tile, j, 2)) m.z = 0.0f;
        if (!in_row(T, tile, j, 3)) m.w = 0.0f;
        s += (m.x * m.x + m.y * m.y) + (m.z * m.z + m.w * m.w);
    }
    s = block_reduce(s, SumOp(), lds + 8);
    const float dn = sqrtf(row_exchange_sum(gran_d + b * C, C, tile, s, base, 2u, spin_limit, lds + 4, &failed));
    if (failed) {
        row_exchange_abandon(fail, b, nullptr, gran_d + b * C, tile, base);
        return;
    }
    const float f = min_nan((1.0f / dn) * eps, 1.0f);
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {      // d recomputed (same expressions, same bits) rather than kept across the exchange
        const float4 x = xs[j * kBlock + threadIdx.x];
        float4 d;
        d.x = (a[j].x + alpha * (g[j].x / gn)) - x.x;
        d.y = (a[j].y + alpha * (g[j].y / gn)) - x.y;
        d.z = (a[j].z + alpha * (g[j].z / gn)) - x.z;
        d.w = (a[j].w + alpha * (g[j].w / gn)) - x.w;
        a[j].x = clampf(x.x + d.x * f, lo, hi);
        a[j].y = clampf(x.y + d.y * f, lo, hi);
        a[j].z = clampf(x.z + d.z * f, lo, hi);
        a[j].w = clampf(x.w + d.w * f, lo, hi);
    }
    store_tile<VEC>(out + b * T, T, tile, a);
    if (tile == 0 && threadIdx.x == 0) {
        if (gnorm) gnorm[b] = gn_raw;
        if (dnorm) dnorm[b] = dn;
    }
}

// The repair kernel queued behind every single-pass launch also closes the call: it moves the row's flag to `last` (diagnostics),
// lowers it, and workgroup 0 advances the workspace's epoch word — after which no granule of this call carries a valid tag.
// Contract (include/advstep.h): the caller zero-fills a workspace once, before its first use (stale FLAGS would send rows to
// the repair pass: slow, not wrong).  Granules need no cleaning and the float partial planes of the other entry points do not
// overlap the exchange area of the same (B, T); a word left by a call of ANOTHER shape is consumed only if it equals this
// call's 32-bit tag at the moment a sibling looks — the epoch makes that a coincidence, not a pattern (the cleaned
// granules of round 4 could meet `last[b] = 1` of another layout, and 1 was a phase tag).
// Returns the row's flag (workgroup-uniform).
__device__ __forceinline__ unsigned take_flag(unsigned *fail, unsigned *last, unsigned *epoch, int64_t b) {
    __shared__ unsigned flag_s;
    if (threadIdx.x == 0) {
        flag_s = fail[b];
        last[b] = flag_s ? kRowFlagged : 0u;
        fail[b] = 0u;
        if (b == 0) *epoch = *epoch + 1u;          // single writer; the next launch on this stream reads it
    }
    __syncthreads();
    return flag_s;
}

// Repair pass of the single-pass step: one workgroup per row, returns at once unless the row's flag is up.  The row's tiles
// are visited in sequence with the three-kernel path's own expressions and reductions (sumsq_partial / pgd_l2_delta /
// pgd_l2_project), so a repaired row carries the same bits as any other.  adv / orig / grad are re-read from global memory:
// the single-pass path is not taken when `out` aliases an input.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void pgd_l2_repair_kernel(const float *__restrict__ adv, const float *__restrict__ grad,
                                                               const float *__restrict__ orig, float *out, int64_t T, int C,
                                                               float alpha, float eps, float eps_div, float lo, float hi,
                                                               unsigned *fail, unsigned *last, unsigned *epoch,
                                                               float *__restrict__ gnorm, float *__restrict__ dnorm) {
    const int64_t b = blockIdx.x;
    if (!take_flag(fail, last, epoch, b)) return;
    __shared__ float gpart[64], dpart[64], lds[12];
    float4 a[kVecs], g[kVecs], x[kVecs];
    for (int tile = 0; tile < C; ++tile) {
        load_tile<VEC>(grad + b * T, T, tile, 0.0f, g);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < kVecs; ++j) s += (g[j].x * g[j].x + g[j].y * g[j].y) + (g[j].z * g[j].z + g[j].w * g[j].w);
        s = block_reduce(s, SumOp(), lds);
        if (threadIdx.x == 0) gpart[tile] = s;
        __syncthreads();
    }
    const float gn_raw = sqrtf(reduce_partials(gpart, C, 0.0f, SumOp(), lds + 4));
    const float gn = gn_raw + eps_div;
    for (int tile = 0; tile < C; ++tile) {
        load_tile<VEC>(adv + b * T, T, tile, 0.0f, a);
        load_tile<VEC>(grad + b * T, T, tile, 0.0f, g);
        load_tile<VEC>(orig + b * T, T, tile, 0.0f, x);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            float4 d;
            d.x = (a[j].x + alpha * (g[j].x / gn)) - x[j].x;
            d.y = (a[j].y + alpha * (g[j].y / gn)) - x[j].y;
            d.z = (a[j].z + alpha * (g[j].z / gn)) - x[j].z;
            d.w = (a[j].w + alpha * (g[j].w / gn)) - x[j].w;
            if (!in_row(T, tile, j, 0)) d.x = 0.0f;
            if (!in_row(T, tile, j, 1)) d.y = 0.0f;
            if (!in_row(T, tile, j, 2)) d.z = 0.0f;
            if (!in_row(T, tile, j, 3)) d.w = 0.0f;
            s += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
        s = block_reduce(s, SumOp(), lds);
        if (threadIdx.x == 0) dpart[tile] = s;
        __syncthreads();
    }
    const float dn = sqrtf(reduce_partials(dpart, C, 0.0f, SumOp(), lds + 8));
    const float f = min_nan((1.0f / dn) * eps, 1.0f);
    for (int tile = 0; tile < C; ++tile) {
        load_tile<VEC>(adv + b * T, T, tile, 0.0f, a);
        load_tile<VEC>(grad + b * T, T, tile, 0.0f, g);
        load_tile<VEC>(orig + b * T, T, tile, 0.0f, x);
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            float4 d;
            d.x = (a[j].x + alpha * (g[j].x / gn)) - x[j].x;
            d.y = (a[j].y + alpha * (g[j].y / gn)) - x[j].y;
            d.z = (a[j].z + alpha * (g[j].z / gn)) - x[j].z;
            d.w = (a[j].w + alpha * (g[j].w / gn)) - x[j].w;
            a[j].x = clampf(x[j].x + d.x * f, lo, hi);
            a[j].y = clampf(x[j].y + d.y * f, lo, hi);
            a[j].z = clampf(x[j].z + d.z * f, lo, hi);
            a[j].w = clampf(x[j].w + d.w * f, lo, hi);
        }
        store_tile<VEC>(out + b * T, T, tile, a);
    }
    if (threadIdx.x == 0) {
        if (gnorm) gnorm[b] = gn_raw;
        if (dnorm) dnorm[b] = dn;
    }
}

// rows whose flag is up (diagnostics / tests: how often the repair pass had work)
__global__ __launch_bounds__(64) void count_flags_kernel(const unsigned *__restrict__ fail, int64_t B, int *__restrict__ count) {
    int n = 0;
    for (int64_t i = threadIdx.x; i < B; i += 64) n += fail[i] == kRowFlagged;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if (threadIdx.x == 0) *count = n;
}

// ---- a6: PGD-L2 random start -----------------------------------------------------------------------------------

// explicit draws: out = clamp(x + normal * ((r / nrm) * eps), lo, hi)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void pgd_l2_init_noise_kernel(const float *__restrict__ x,
                                                                   const float *__restrict__ normal,
                                                                   const float *__restrict__ r, float *out, int64_t T,
                                                                   float eps, float lo, float hi,
                                                                   const float *__restrict__ npart) {
    __shared__ float lds[4];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 xv[kVecs], nz[kVecs];
    load_tile<VEC>(x + b * T, T, tile, 0.0f, xv);
    load_tile<VEC>(normal + b * T, T, tile, 0.0f, nz);
    const float nrm = sqrtf(reduce_partials(npart + b * C, C, 0.0f, SumOp(), lds));
    const float scale = (r[b] / nrm) * eps;
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        xv[j].x = clampf(xv[j].x + nz[j].x * scale, lo, hi);
        xv[j].y = clampf(xv[j].y + nz[j].y * scale, lo, hi);
        xv[j].z = clampf(xv[j].z + nz[j].z * scale, lo, hi);
        xv[j].w = clampf(xv[j].w + nz[j].w * scale, lo, hi);
    }
    store_tile<VEC>(out + b * T, T, tile, xv);
}

// Philox normals for this thread's kVecs quads of (row b, tile); out-of-row samples are zeroed.
__device__ __forceinline__ void philox_normal_tile(int64_t T, int tile, uint32_t b, uint64_t seed, uint64_t offset,
                                                   float4 (&nz)[kVecs]) {
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        const int64_t q = (int64_t)tile * kTileVec + j * kBlock + threadIdx.x;
        if (q * 4 < T) {
            nz[j] = philox_normal4((uint32_t)q, b, seed, offset);
            if (q * 4 + 1 >= T) nz[j].y = 0.0f;
            if (q * 4 + 2 >= T) nz[j].z = 0.0f;
            if (q * 4 + 3 >= T) nz[j].w = 0.0f;
        } else {
            nz[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
}

__global__ __launch_bounds__(kBlock) void philox_normal_sumsq_kernel(int64_t T, uint64_t seed, uint64_t offset,
                                                                     float *__restrict__ part) {
    __shared__ float lds[4];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 nz[kVecs];
    philox_normal_tile(T, tile, (uint32_t)b, seed, offset, nz);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kVecs; ++j)
        s += (nz[j].x * nz[j].x + nz[j].y * nz[j].y) + (nz[j].z * nz[j].z + nz[j].w * nz[j].w);
    s = block_reduce(s, SumOp(), lds);
    if (threadIdx.x == 0) part[b * C + tile] = s;
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void pgd_l2_init_philox_kernel(const float *__restrict__ x, float *out,
                                                                    int64_t T, float eps, float lo, float hi,
                                                                    uint64_t seed, uint64_t offset,
                                                                    const float *__restrict__ npart) {
    __shared__ float lds[4];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 xv[kVecs], nz[kVecs];
    load_tile<VEC>(x + b * T, T, tile, 0.0f, xv);
    philox_normal_tile(T, tile, (uint32_t)b, seed, offset, nz);
    const float nrm = sqrtf(reduce_partials(npart + b * C, C, 0.0f, SumOp(), lds));
    const uint64_t off1 = offset + 1;
    const Quad rq = philox4x32_10((uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)off1, (uint32_t)(off1 >> 32),
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const float scale = (u01(rq.v[0]) / nrm) * eps;
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        xv[j].x = clampf(xv[j].x + nz[j].x * scale, lo, hi);
        xv[j].y = clampf(xv[j].y + nz[j].y * scale, lo, hi);
        xv[j].z = clampf(xv[j].z + nz[j].z * scale, lo, hi);
        xv[j].w = clampf(xv[j].w + nz[j].w * scale, lo, hi);
    }
    store_tile<VEC>(out + b * T, T, tile, xv);
}

// The same start in ONE launch: the normals are generated once and stay in registers across the row exchange of ||n||^2
// (granule protocol of pgd_l2_fused_kernel; same partial sums and re-reduction as the two-kernel path: bit-identical).
template <bool VEC>
__global__ __launch_bounds__(kBlock, 8) void pgd_l2_init_philox_fused_kernel(const float *__restrict__ x, float *out, int64_t T,
                                                                             float eps, float lo, float hi, uint64_t seed,
                                                                             uint64_t offset,
                                                                             unsigned long long *__restrict__ gran,
                                                                             unsigned *__restrict__ fail,
                                                                             const unsigned *__restrict__ epoch,
                                                                             unsigned spin_limit) {
    __shared__ float lds[8];
    __shared__ int failed;
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 nz[kVecs];
    philox_normal_tile(T, tile, (uint32_t)b, seed, offset, nz);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kVecs; ++j)
        s += (nz[j].x * nz[j].x + nz[j].y * nz[j].y) + (nz[j].z * nz[j].z + nz[j].w * nz[j].w);
    s = block_reduce(s, SumOp(), lds);
    // the tile of x is requested BEFORE the exchange (16-byte path): its HBM latency then runs under the wait for the row's other
    // workgroups; the scalar path (rows not 16-byte addressable) keeps it behind - 16 more live registers make it spill
    float4 xv[kVecs];
    if constexpr (VEC) load_tile<VEC>(x + b * T, T, tile, 0.0f, xv);
    const unsigned base = exchange_tag_base(epoch);
    const float nrm = sqrtf(row_exchange_sum(gran + b * C, C, tile, s, base, 1u, spin_limit, lds + 4, &failed));
    if (failed) {
        row_exchange_abandon(fail, b, gran + b * C, nullptr, tile, base);
        return;
    }
    const uint64_t off1 = offset + 1;
    const Quad rq = philox4x32_10((uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)off1, (uint32_t)(off1 >> 32),
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const float scale = (u01(rq.v[0]) / nrm) * eps;
    if constexpr (!VEC) load_tile<VEC>(x + b * T, T, tile, 0.0f, xv);
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        xv[j].x = clampf(xv[j].x + nz[j].x * scale, lo, hi);
        xv[j].y = clampf(xv[j].y + nz[j].y * scale, lo, hi);
        xv[j].z = clampf(xv[j].z + nz[j].z * scale, lo, hi);
        xv[j].w = clampf(xv[j].w + nz[j].w * scale, lo, hi);
    }
    store_tile<VEC>(out + b * T, T, tile, xv);
}

// Repair pass of the single-pass start (see pgd_l2_repair_kernel): the two-kernel path's arithmetic, one workgroup per
// flagged row.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void pgd_l2_init_philox_repair_kernel(const float *__restrict__ x, float *out, int64_t T, int C,
                                                                           float eps, float lo, float hi, uint64_t seed,
                                                                           uint64_t offset, unsigned *fail, unsigned *last,
                                                                           unsigned *epoch) {
    const int64_t b = blockIdx.x;
    if (!take_flag(fail, last, epoch, b)) return;
    __shared__ float npart[64], lds[8];
    float4 nz[kVecs], xv[kVecs];
    for (int tile = 0; tile < C; ++tile) {
        philox_normal_tile(T, tile, (uint32_t)b, seed, offset, nz);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < kVecs; ++j)
            s += (nz[j].x * nz[j].x + nz[j].y * nz[j].y) + (nz[j].z * nz[j].z + nz[j].w * nz[j].w);
        s = block_reduce(s, SumOp(), lds);
        if (threadIdx.x == 0) npart[tile] = s;
        __syncthreads();
    }
    const float nrm = sqrtf(reduce_partials(npart, C, 0.0f, SumOp(), lds + 4));
    const uint64_t off1 = offset + 1;
    const Quad rq = philox4x32_10((uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)off1, (uint32_t)(off1 >> 32),
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const float scale = (u01(rq.v[0]) / nrm) * eps;
    for (int tile = 0; tile < C; ++tile) {
        load_tile<VEC>(x + b * T, T, tile, 0.0f, xv);
        philox_normal_tile(T, tile, (uint32_t)b, seed, offset, nz);
#pragma unroll
        for (int j = 0; j < kVecs; ++j) {
            xv[j].x = clampf(xv[j].x + nz[j].x * scale, lo, hi);
            xv[j].y = clampf(xv[j].y + nz[j].y * scale, lo, hi);
            xv[j].z = clampf(xv[j].z + nz[j].z * scale, lo, hi);
            xv[j].w = clampf(xv[j].w + nz[j].w * scale, lo, hi);
        }
        store_tile<VEC>(out + b * T, T, tile, xv);
    }
}

// ---- a7: CW -------------------------------------------------------------------------------------------------------

// adv = 1/2 * (tanh(w) + 1); partial sum (adv - x)^2
template <bool VEC>
__global__ __launch_bounds__(kBlock) void cw_tanh_sqdist_kernel(const float *__restrict__ w,
                                                                const float *__restrict__ x, float *adv, int64_t T,
                                                                float *__restrict__ part) {
    __shared__ float lds[4];
    const int tile = blockIdx.x, C = gridDim.x;
    const int64_t b = blockIdx.y;
    float4 wv[kVecs], xv[kVecs];
    load_tile<VEC>(w + b * T, T, tile, 0.0f, wv);
    // fill = 0.5 == tanh_space(0): out-of-row lanes contribute (0.5 - 0.5)^2 = 0
    load_tile<VEC>(x + b * T, T, tile, 0.5f, xv);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        wv[j].x = 0.5f * (tanhf(wv[j].x) + 1.0f);
        wv[j].y = 0.5f * (tanhf(wv[j].y) + 1.0f);
        wv[j].z = 0.5f * (tanhf(wv[j].z) + 1.0f);
        wv[j].w = 0.5f * (tanhf(wv[j].w) + 1.0f);
        const float dx = wv[j].x - xv[j].x, dy = wv[j].y - xv[j].y, dz = wv[j].z - xv[j].z, dw = wv[j].w - xv[j].w;
        s += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    store_tile<VEC>(adv + b * T, T, tile, wv);
    s = block_reduce(s, SumOp(), lds);
    if (threadIdx.x == 0) part[b * C + tile] = s;
}

// one wave per row: l2[b] = sum of the row's partials
__global__ __launch_bounds__(64) void row_partials_sum_kernel(const float *__restrict__ part, int C,
                                                              float *__restrict__ out) {
    const int64_t b = blockIdx.x;
    float v = 0.0f;
    for (int i = threadIdx.x; i < C; i += 64) v += part[b * C + i];
    v = wave_reduce(v, SumOp());
    if (threadIdx.x == 0) out[b] = v;
}

// best = mask * adv + (1 - mask) * best
template <bool VEC>
__global__ __launch_bounds__(kBlock) void cw_best_update_kernel(const float *__restrict__ adv,
                                                                const float *__restrict__ mask, float *best,
                                                                int64_t T) {
    const int tile = blockIdx.x;
    const int64_t b = blockIdx.y;
    const float mk = mask[b];
    const float inv = 1.0f - mk;
    float4 a[kVecs], bs[kVecs];
    load_tile<VEC>(adv + b * T, T, tile, 0.0f, a);
    load_tile<VEC>(best + b * T, T, tile, 0.0f, bs);
#pragma unroll
    for (int j = 0; j < kVecs; ++j) {
        bs[j].x = mk * a[j].x + inv * bs[j].x;
        bs[j].y = mk * a[j].y + inv * bs[j].y;
        bs[j].z = mk * a[j].z + inv * bs[j].z;
        bs[j].w = mk * a[j].w + inv * bs[j].w;
    }
    store_tile<VEC>(best + b * T, T, tile, bs);
}

// ---- a8: 2-logit cross-entropy, closed form ------------------------------------------------------------------------

__device__ __forceinline__ float softplusf(float t) {  // log(1 + exp(t)), stable
    return (t > 0.0f ? t : 0.0f) + log1pf(expf(-fabsf(t)));
}

__global__ __launch_bounds__(kBlock) void ce2_loss_grad_kernel(const float *__restrict__ z,
                                                               const int64_t *__restrict__ labels,
                                                               float *__restrict__ dz, float *__restrict__ loss,
                                                               int64_t B, float scale) {
    __shared__ float lds[4];
    const float invB = 1.0f / (float)B;
    float acc = 0.0f;
    for (int64_t b = threadIdx.x; b < B; b += kBlock) {
        // CE([-z, z], y) = softplus(u), u = (1 - 2y) * 2z;  d/dz = (1 - 2y) * 2 * sigmoid(u).
        // (sigmoid(2z) - y written without the cancellation at saturated logits.)
        const float flip = 1.0f - 2.0f * (float)labels[b];
        const float u = flip * (2.0f * z[b]);
        acc += softplusf(u);
        const float sig = 1.0f / (1.0f + expf(-u));
        dz[b] = scale * ((2.0f * invB) * (flip * sig));
    }
    acc = block_reduce(acc, SumOp(), lds);
    if (threadIdx.x == 0) loss[0] = scale * (acc * invB);
}

// ---------------------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------------------

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int tiles_per_row(int64_t T) { return (int)ceil_div(T, kTile); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

struct RowWs {
    float *p0;
    float *p1;
    unsigned *epoch;               // single-pass PGD-L2 exchange: call counter (the tags' epoch) ...
    unsigned long long *gran0;     // ... two planes of B x C 8-byte {tag, value} granules ...
    unsigned long long *gran1;
    unsigned *fail;                // ... the live "row needs repair" flags and the flags of the LAST single-pass call as its
    unsigned *last;                //     repair pass left them (advstep_pgd_l2_repaired_rows reads those)
};
// Layout for a (B, T) batch, C = tiles per row (round 5: the exchange area no longer lies over the float planes):
//   [16-byte header: epoch word][granule plane 0][granule plane 1][fail flags][last flags][float plane 0][float plane 1]
inline size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }
constexpr size_t kWsHeader = 16;
inline size_t row_ws_plane(int64_t B, int64_t T) { return align16((size_t)B * (size_t)tiles_per_row(T) * sizeof(float)); }
inline size_t row_ws_granules(int64_t B, int64_t T) {
    return align16((size_t)B * (size_t)tiles_per_row(T) * sizeof(unsigned long long));
}
inline size_t row_ws_flags(int64_t B) { return align16((size_t)B * sizeof(unsigned)); }
inline size_t row_ws_bytes(int64_t B, int64_t T) {
    return kWsHeader + 2 * row_ws_granules(B, T) + 2 * row_ws_flags(B) + 2 * row_ws_plane(B, T);
}
inline bool carve_ws(void *ws, size_t ws_bytes, int64_t B, int64_t T, RowWs *out) {
    if (!ws || !aligned16(ws) || ws_bytes < row_ws_bytes(B, T)) return false;
    char *p = static_cast<char *>(ws);
    out->epoch = reinterpret_cast<unsigned *>(p);
    p += kWsHeader;
    out->gran0 = reinterpret_cast<unsigned long long *>(p);
    p += row_ws_granules(B, T);
    out->gran1 = reinterpret_cast<unsigned long long *>(p);
    p += row_ws_granules(B, T);
    out->fail = reinterpret_cast<unsigned *>(p);
    p += row_ws_flags(B);
    out->last = reinterpret_cast<unsigned *>(p);
    p += row_ws_flags(B);
    out->p0 = reinterpret_cast<float *>(p);
    out->p1 = reinterpret_cast<float *>(p + row_ws_plane(B, T));
    return true;
}

// ADVSTEP_L2_SINGLE_PASS=0 keeps the three-kernel PGD-L2 step (A/B measurements, read at every call); default on.
inline bool l2_single_pass() {
    const char *e = getenv("ADVSTEP_L2_SINGLE_PASS");
    return !(e && e[0] == '0');
}

// Bound of the in-launch wait, in sweeps (one sweep = an agent-scope load + s_sleep, ~1 us): ~0.1 s, after which the row goes
// to the repair pass.  ADVSTEP_L2_SPIN_LIMIT overrides it (0 = every row gives up at once: tests of the repair pass).
inline unsigned l2_spin_limit() {
    const char *e = getenv("ADVSTEP_L2_SPIN_LIMIT");
    if (e && *e) return (unsigned)strtoul(e, nullptr, 10);
    return 1u << 17;
}

// How many workgroups of `kernel` (kBlock threads, its static LDS) the CURRENT device holds at once: CU count x the
// occupancy the runtime computes for this kernel's registers / LDS.  Queried once per (device, kernel); 0 if the query
// fails (the caller then takes the multi-kernel path).  ADVSTEP_L2_CAPACITY overrides it (tests: a "smaller device").
enum { kCapStepVec, kCapStepScalar, kCapInitVec, kCapInitScalar, kCapKinds };
inline int64_t l2_resident_capacity(int kind, const void *kernel) {
    const char *e = getenv("ADVSTEP_L2_CAPACITY");
    if (e && *e) return (int64_t)strtoll(e, nullptr, 10);
    static std::atomic<int64_t> cache[kCapKinds][64];      // zero-initialised: 0 = not queried yet, -1 = query failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int64_t c = cache[kind][dev].load(std::memory_order_relaxed);
    if (c == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) == hipSuccess && cus > 0 && per_cu > 0)
            c = (int64_t)cus * per_cu;
        else
            c = -1;
        (void)hipGetLastError();
        cache[kind][dev].store(c, std::memory_order_relaxed);
    }
    return c > 0 ? c : 0;
}

inline bool overlaps(const void *a, const void *b, size_t bytes) {
    const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
    return x < y + bytes && y < x + bytes;
}

// grid.y is limited to 65535 rows per launch; batches beyond that are launched in slabs.
constexpr int64_t kMaxRowsPerLaunch = 65535;

// float4 per thread per stream of the flat kernels.  Measured on MI355X at B = 128, T = 64 600 (tools/tune_pgd_step.hip,
// profiles/): one float4 per stream per thread (8 076 workgroups) beats 4 (2 019 workgroups) by 3 % hot / 7 % cold.
// ADVSTEP_FLAT_VECS = 1 | 2 | 4 overrides it for experiments.
inline int flat_vecs() {
    static const int v = [] {
        const char *e = getenv("ADVSTEP_FLAT_VECS");
        const int x = e ? atoi(e) : 1;
        return (x == 1 || x == 2 || x == 4) ? x : 1;
    }();
    return v;
}

template <int NIN, int VECS, class Op>
void launch_flat_vec(const float *in0, const float *in1, const float *in2, float *out, int64_t n4, Op op,
                     hipStream_t st) {
    const int64_t ntiles = ceil_div(n4, kBlock * VECS);
    const int64_t cap = (int64_t)kMaxGrid * (4 / VECS);
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    hipLaunchKernelGGL((flat_vec_kernel<NIN, VECS, Op>), dim3(grid), dim3(kBlock), 0, st,
                       reinterpret_cast<const float4 *>(in0), reinterpret_cast<const float4 *>(in1),
                       reinterpret_cast<const float4 *>(in2), reinterpret_cast<float4 *>(out), n4, ntiles, op);
}

template <int NIN, class Op>
int launch_flat(const float *in0, const float *in1, const float *in2, float *out, int64_t n, Op op,
                hipStream_t st) {
    if (n == 0) return ADVSTEP_OK;
    const bool vec = aligned16(in0) && aligned16(out) && (NIN < 2 || aligned16(in1)) && (NIN < 3 || aligned16(in2));
    const int64_t n4 = vec ? n / 4 : 0;
    if (n4 > 0) {
        switch (flat_vecs()) {
            case 4: launch_flat_vec<NIN, 4>(in0, in1, in2, out, n4, op, st); break;
            case 2: launch_flat_vec<NIN, 2>(in0, in1, in2, out, n4, op, st); break;
            default: launch_flat_vec<NIN, 1>(in0, in1, in2, out, n4, op, st); break;
        }
    }
    const int64_t begin = n4 * 4;
    if (begin < n) {
        const int64_t blocks = ceil_div(n - begin, kBlock);
        const int grid = (int)(blocks < kMaxGrid ? blocks : kMaxGrid);
        hipLaunchKernelGGL((flat_scalar_kernel<NIN, Op>), dim3(grid), dim3(kBlock), 0, st, in0, in1, in2, out, begin,
                           n, op);
    }
    return status_after_launch();
}

inline bool rows_vec(int64_t T, std::initializer_list<const void *> ptrs) {
    if (T % 4 != 0) return false;
    for (const void *p : ptrs)
        if (!aligned16(p)) return false;
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------

#define ADV_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

// Launch a row kernel over all B rows in slabs of <= 65535 rows; `ARGS` may use `b0` (first row of the slab).
#define ADV_LAUNCH_ROWS(KERNEL, VECFLAG, B, T, st, ...)                                          \
    do {                                                                                         \
        const int C_ = tiles_per_row(T);                                                         \
        for (int64_t b0 = 0; b0 < (B); b0 += kMaxRowsPerLaunch) {                                \
            const int64_t nb_ = ((B)-b0 < kMaxRowsPerLaunch) ? ((B)-b0) : kMaxRowsPerLaunch;     \
            if (VECFLAG)                                                                         \
                hipLaunchKernelGGL((KERNEL<true>), dim3(C_, (unsigned)nb_), dim3(kBlock), 0, st, __VA_ARGS__); \
            else                                                                                 \
                hipLaunchKernelGGL((KERNEL<false>), dim3(C_, (unsigned)nb_), dim3(kBlock), 0, st, __VA_ARGS__); \
        }                                                                                        \
    } while (0)

extern "C" {

int advstep_abi_version(void) { return ADVSTEP_ABI_VERSION; }

const char *advstep_status_string(int status) {
    switch (status) {
        case ADVSTEP_OK: return "ok";
        case ADVSTEP_EINVAL: return "invalid argument";
        case ADVSTEP_EWORKSPACE: return "row workspace missing, misaligned or too small";
        case ADVSTEP_ELAUNCH: return "HIP kernel launch failed";
        case ADVSTEP_ENODEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

int advstep_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

size_t advstep_row_workspace_bytes(int64_t B, int64_t T) {
    if (B <= 0 || T <= 0) return 0;
    return row_ws_bytes(B, T);
}

int advstep_minmax_normalize_f32(const float *x, float *x01, float *mn, float *mx, int64_t B, int64_t T, void *ws,
                                 size_t ws_bytes, advstep_stream_t stream) {
    ADV_REQUIRE(B >= 0 && T >= 0);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    ADV_REQUIRE(x && x01 && mn && mx && x != x01);
    RowWs w;
    if (!carve_ws(ws, ws_bytes, B, T, &w)) return ADVSTEP_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const bool vec = rows_vec(T, {x, x01});
    const int C = tiles_per_row(T);
    ADV_LAUNCH_ROWS(minmax_partial_kernel, vec, B, T, st, x + b0 * T, T, w.p0 + b0 * C, w.p1 + b0 * C);
    ADV_LAUNCH_ROWS(minmax_apply_kernel, vec, B, T, st, x + b0 * T, x01 + b0 * T, mn + b0, mx + b0, T, w.p0 + b0 * C,
                    w.p1 + b0 * C);
    return status_after_launch();
}

int advstep_minmax_revert_f32(const float *x01, const float *mn, const float *mx, float *out, int64_t B, int64_t T,
                              advstep_stream_t stream) {
    ADV_REQUIRE(B >= 0 && T >= 0);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    ADV_REQUIRE(x01 && mn && mx && out);
    hipStream_t st = as_stream(stream);
    const bool vec = rows_vec(T, {x01, out});
    ADV_LAUNCH_ROWS(minmax_revert_kernel, vec, B, T, st, x01 + b0 * T, mn + b0, mx + b0, out + b0 * T, T);
    return status_after_launch();
}

int advstep_fgsm_step_f32(const float *x, const float *grad, float *out, int64_t n, float eps, float lo, float hi,
                          advstep_stream_t stream) {
    ADV_REQUIRE(n >= 0);
    if (n == 0) return ADVSTEP_OK;
    ADV_REQUIRE(x && grad && out);
    return launch_flat<2>(x, grad, nullptr, out, n, FgsmOp{eps, lo, hi}, as_stream(stream));
}

int advstep_pgd_linf_init_noise_f32(const float *x, const float *noise, float *out, int64_t n, float lo, float hi,
                                    advstep_stream_t stream) {
    ADV_REQUIRE(n >= 0);
    if (n == 0) return ADVSTEP_OK;
    ADV_REQUIRE(x && noise && out);
    return launch_flat<2>(x, noise, nullptr, out, n, AddClampOp{lo, hi}, as_stream(stream));
}

int advstep_pgd_linf_init_philox_f32(const float *x, float *out, int64_t n, float eps, float lo, float hi,
                                     uint64_t seed, uint64_t offset, advstep_stream_t stream) {
    ADV_REQUIRE(n >= 0);
    if (n == 0) return ADVSTEP_OK;
    ADV_REQUIRE(x && out);
    hipStream_t st = as_stream(stream);
    const bool vec = aligned16(x) && aligned16(out);
    const int64_t n4 = vec ? n / 4 : 0;
    if (n4 > 0) {
        const int64_t ntiles = ceil_div(n4, kTileVec);
        const int grid = (int)(ntiles < kMaxGrid ? ntiles : kMaxGrid);
        hipLaunchKernelGGL(pgd_linf_init_philox_vec_kernel, dim3(grid), dim3(kBlock), 0, st,
                           reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(out), n4, ntiles, eps, lo,
                           hi, seed, offset);
    }
    const int64_t begin = n4 * 4;
    if (begin < n) {
        const int64_t blocks = ceil_div(n - begin, kBlock);
        const int grid = (int)(blocks < kMaxGrid ? blocks : kMaxGrid);
        hipLaunchKernelGGL(pgd_linf_init_philox_scalar_kernel, dim3(grid), dim3(kBlock), 0, st, x, out, begin, n, eps,
                           lo, hi, seed, offset);
    }
    return status_after_launch();
}

int advstep_pgd_linf_step_f32(const float *adv, const float *grad, const float *orig, float *out, int64_t n,
                              float alpha, float eps, float lo, float hi, advstep_stream_t stream) {
    ADV_REQUIRE(n >= 0);
    if (n == 0) return ADVSTEP_OK;
    ADV_REQUIRE(adv && grad && orig && out);
    return launch_flat<3>(adv, grad, orig, out, n, PgdLinfOp{alpha, eps, lo, hi}, as_stream(stream));
}

int advstep_pgd_l2_init_noise_f32(const float *x, const float *normal, const float *r, float *out, int64_t B,
                                  int64_t T, float eps, float lo, float hi, void *ws, size_t ws_bytes,
                                  advstep_stream_t stream) {
    ADV_REQUIRE(B >= 0 && T >= 0);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    ADV_REQUIRE(x && normal && r && out);
    RowWs w;
    if (!carve_ws(ws, ws_bytes, B, T, &w)) return ADVSTEP_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const bool vec = rows_vec(T, {x, normal, out});
    const int C = tiles_per_row(T);
    ADV_LAUNCH_ROWS(sumsq_partial_kernel, vec, B, T, st, normal + b0 * T, T, w.p0 + b0 * C);
    ADV_LAUNCH_ROWS(pgd_l2_init_noise_kernel, vec, B, T, st, x + b0 * T, normal + b0 * T, r + b0, out + b0 * T, T, eps,
                    lo, hi, w.p0 + b0 * C);
    return status_after_launch();
}

int advstep_pgd_l2_init_philox_f32(const float *x, float *out, int64_t B, int64_t T, float eps, float lo, float hi,
                                   uint64_t seed, uint64_t offset, void *ws, size_t ws_bytes,
                                   advstep_stream_t stream) {
    ADV_REQUIRE(B >= 0 && T >= 0);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    ADV_REQUIRE(x && out);
    ADV_REQUIRE(B <= kMaxRowsPerLaunch);  // the Philox counter carries the absolute row index
    RowWs w;
    if (!carve_ws(ws, ws_bytes, B, T, &w)) return ADVSTEP_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const bool vec = rows_vec(T, {x, out});
    const int C = tiles_per_row(T);
    const void *fused = vec ? (const void *)pgd_l2_init_philox_fused_kernel<true> : (const void *)pgd_l2_init_philox_fused_kernel<false>;
    if (l2_single_pass() && C <= 64 && !overlaps(x, out, (size_t)B * T * sizeof(float)) &&
        B * C <= l2_resident_capacity(vec ? kCapInitVec : kCapInitScalar, fused)) {
        // the repair kernel behind the launch serves flagged rows, lowers their flags and advances the epoch (closes the call)
        const unsigned spins = l2_spin_limit();
        if (vec) {
            hipLaunchKernelGGL(pgd_l2_init_philox_fused_kernel<true>, dim3(C, (unsigned)B), dim3(kBlock), 0, st, x, out, T, eps, lo,
                               hi, seed, offset, w.gran0, w.fail, w.epoch, spins);
            hipLaunchKernelGGL(pgd_l2_init_philox_repair_kernel<true>, dim3((unsigned)B), dim3(kBlock), 0, st, x, out, T, C, eps,
                               lo, hi, seed, offset, w.fail, w.last, w.epoch);
        } else {
            hipLaunchKernelGGL(pgd_l2_init_philox_fused_kernel<false>, dim3(C, (unsigned)B), dim3(kBlock), 0, st, x, out, T, eps, lo,
                               hi, seed, offset, w.gran0, w.fail, w.epoch, spins);
            hipLaunchKernelGGL(pgd_l2_init_philox_repair_kernel<false>, dim3((unsigned)B), dim3(kBlock), 0, st, x, out, T, C, eps,
                               lo, hi, seed, offset, w.fail, w.last, w.epoch);
        }
        return status_after_launch();
    }
    hipLaunchKernelGGL(philox_normal_sumsq_kernel, dim3(C, (unsigned)B), dim3(kBlock), 0, st, T, seed, offset, w.p0);
    ADV_LAUNCH_ROWS(pgd_l2_init_philox_kernel, vec, B, T, st, x, out, T, eps, lo, hi, seed, offset, w.p0);
    return status_after_launch();
}

int advstep_pgd_l2_step_f32(const float *adv, const float *grad, const float *orig, float *out, int64_t B, int64_t T,
                            float alpha, float eps, float eps_div, float lo, float hi, float *gnorm, float *dnorm,
                            void *ws, size_t ws_bytes, advstep_stream_t stream) {
    ADV_REQUIRE(B >= 0 && T >= 0);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    ADV_REQUIRE(adv && grad && orig && out);
    RowWs w;
    if (!carve_ws(ws, ws_bytes, B, T, &w)) return ADVSTEP_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const bool vec = rows_vec(T, {adv, grad, orig, out});
    const int C = tiles_per_row(T);
    const void *fused = vec ? (const void *)pgd_l2_fused_kernel<true> : (const void *)pgd_l2_fused_kernel<false>;
    const size_t row_bytes = (size_t)B * T * sizeof(float);
    if (l2_single_pass() && C <= 64 && !overlaps(adv, out, row_bytes) && !overlaps(grad, out, row_bytes) &&
        !overlaps(orig, out, row_bytes) && B * C <= l2_resident_capacity(vec ? kCapStepVec : kCapStepScalar, fused)) {
        // the launch fits the device's resident capacity for this kernel: one launch, 16 B per sample; the repair kernel behind
        // it serves rows whose exchange was abandoned, lowers their flags and advances the epoch (no memset node, nothing cleaned)
        const unsigned spins = l2_spin_limit();
        if (vec) {
            hipLaunchKernelGGL(pgd_l2_fused_kernel<true>, dim3(C, (unsigned)B), dim3(kBlock), 0, st, adv, grad, orig, out, T, alpha,
                               eps, eps_div, lo, hi, w.gran0, w.gran1, w.fail, w.epoch, spins, gnorm, dnorm);
            hipLaunchKernelGGL(pgd_l2_repair_kernel<true>, dim3((unsigned)B), dim3(kBlock), 0, st, adv, grad, orig, out, T, C, alpha,
                               eps, eps_div, lo, hi, w.fail, w.last, w.epoch, gnorm, dnorm);
        } else {
            hipLaunchKernelGGL(pgd_l2_fused_kernel<false>, dim3(C, (unsigned)B), dim3(kBlock), 0, st, adv, grad, orig, out, T, alpha,
                               eps, eps_div, lo, hi, w.gran0, w.gran1, w.fail, w.epoch, spins, gnorm, dnorm);
            hipLaunchKernelGGL(pgd_l2_repair_kernel<false>, dim3((unsigned)B), dim3(kBlock), 0, st, adv, grad, orig, out, T, C, alpha,
                               eps, eps_div, lo, hi, w.fail, w.last, w.epoch, gnorm, dnorm);
        }
        return status_after_launch();
    }
    ADV_LAUNCH_ROWS(sumsq_partial_kernel, vec, B, T, st, grad + b0 * T, T, w.p0 + b0 * C);
    ADV_LAUNCH_ROWS(pgd_l2_delta_kernel, vec, B, T, st, adv + b0 * T, grad + b0 * T, orig + b0 * T, T, alpha, eps_div,
                    w.p0 + b0 * C, w.p1 + b0 * C, gnorm ? gnorm + b0 : nullptr);
    ADV_LAUNCH_ROWS(pgd_l2_project_kernel, vec, B, T, st, adv + b0 * T, grad + b0 * T, orig + b0 * T, out + b0 * T, T,
                    alpha, eps, eps_div, lo, hi, w.p0 + b0 * C, w.p1 + b0 * C, dnorm ? dnorm + b0 : nullptr);
    return status_after_launch();
}

int advstep_pgd_l2_repaired_rows(const void *ws, size_t ws_bytes, int64_t B, int64_t T, int *count,
                                 advstep_stream_t stream) {
    ADV_REQUIRE(B >= 0 && T >= 0 && count);
    if (B == 0 || T == 0) return hipMemsetAsync(count, 0, sizeof(int), as_stream(stream)) == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH;
    if (!ws || !aligned16(ws) || ws_bytes < row_ws_bytes(B, T)) return ADVSTEP_EWORKSPACE;
    RowWs w;
    carve_ws(const_cast<void *>(ws), ws_bytes, B, T, &w);
    hipLaunchKernelGGL(count_flags_kernel, dim3(1), dim3(64), 0, as_stream(stream), (const unsigned *)w.last, B, count);
    return status_after_launch();
}

int advstep_cw_init_w_f32(const float *x, float *w, int64_t n, advstep_stream_t stream) {
    ADV_REQUIRE(n >= 0);
    if (n == 0) return ADVSTEP_OK;
    ADV_REQUIRE(x && w);
    return launch_flat<1>(x, nullptr, nullptr, w, n, CwInitOp{}, as_stream(stream));
}

int advstep_cw_tanh_sqdist_f32(const float *w, const float *x, float *adv, float *l2, int64_t B, int64_t T, void *ws,
                               size_t ws_bytes, advstep_stream_t stream) {
    ADV_REQUIRE(B >= 0 && T >= 0);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    ADV_REQUIRE(w && x && adv && l2);
    RowWs wsp;
    if (!carve_ws(ws, ws_bytes, B, T, &wsp)) return ADVSTEP_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const bool vec = rows_vec(T, {w, x, adv});
    const int C = tiles_per_row(T);
    ADV_LAUNCH_ROWS(cw_tanh_sqdist_kernel, vec, B, T, st, w + b0 * T, x + b0 * T, adv + b0 * T, T, wsp.p0 + b0 * C);
    hipLaunchKernelGGL(row_partials_sum_kernel, dim3((unsigned)B), dim3(64), 0, st, wsp.p0, C, l2);
    return status_after_launch();
}

int advstep_cw_adam_step_f32(float *w, float *m, float *v, const float *x, const float *grad_adv, int64_t n,
                             int64_t step, double lr, double beta1, double beta2, double adam_eps,
                             advstep_stream_t stream) {
    ADV_REQUIRE(n >= 0 && step >= 1);
    if (n == 0) return ADVSTEP_OK;
    ADV_REQUIRE(w && m && v && x && grad_adv);
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars s;
    s.w1 = (float)(1.0 - beta1);
    s.beta2 = (float)beta2;
    s.omb2 = (float)(1.0 - beta2);
    s.neg_step = (float)(-(lr / bc1));
    s.bc2_sqrt = (float)sqrt(bc2);
    s.eps = (float)adam_eps;
    hipStream_t st = as_stream(stream);
    const bool vec = aligned16(w) && aligned16(m) && aligned16(v) && aligned16(x) && aligned16(grad_adv);
    const int64_t n4 = vec ? n / 4 : 0;
    if (n4 > 0) {
        const int64_t ntiles = ceil_div(n4, kTileVec);
        const int grid = (int)(ntiles < kMaxGrid ? ntiles : kMaxGrid);
        hipLaunchKernelGGL(cw_adam_vec_kernel, dim3(grid), dim3(kBlock), 0, st, reinterpret_cast<float4 *>(w),
                           reinterpret_cast<float4 *>(m), reinterpret_cast<float4 *>(v),
                           reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(grad_adv), n4, ntiles,
                           s);
    }
    const int64_t begin = n4 * 4;
    if (begin < n) {
        const int64_t blocks = ceil_div(n - begin, kBlock);
        const int grid = (int)(blocks < kMaxGrid ? blocks : kMaxGrid);
        hipLaunchKernelGGL(cw_adam_scalar_kernel, dim3(grid), dim3(kBlock), 0, st, w, m, v, x, grad_adv, begin, n, s);
    }
    return status_after_launch();
}

int advstep_cw_best_update_f32(const float *adv, const float *mask, float *best, int64_t B, int64_t T,
                               advstep_stream_t stream) {
    ADV_REQUIRE(B >= 0 && T >= 0);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    ADV_REQUIRE(adv && mask && best);
    hipStream_t st = as_stream(stream);
    const bool vec = rows_vec(T, {adv, best});
    ADV_LAUNCH_ROWS(cw_best_update_kernel, vec, B, T, st, adv + b0 * T, mask + b0, best + b0 * T, T);
    return status_after_launch();
}

int advstep_ce2_loss_grad_f32(const float *z, const int64_t *labels, float *dz, float *loss, int64_t B, float scale,
                              advstep_stream_t stream) {
    ADV_REQUIRE(B >= 1);
    ADV_REQUIRE(z && labels && dz && loss);
    hipLaunchKernelGGL(ce2_loss_grad_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), z, labels, dz, loss, B,
                       scale);
    return status_after_launch();
}

}  // extern "C"
