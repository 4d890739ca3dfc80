// fab.hip — the FAB attack's per-iteration tensor work on gfx950 (C ABI: include/advstep_fab.h).
//
// Reference: adversarial_attacks/torchattacks/attacks/fab.py:208-292 (loop body), :562-717 (projections).
//
// Layout: every kernel gives ONE workgroup of 1024 threads (16 wave64) to ONE row of T samples.  At the repo's
// T = 64 600 a row is 258 KB: it does not fit LDS, but the (t, w) pair of a row stays L2 / Infinity-Cache resident
// between the passes its workgroup makes over it, so only the first pass reads HBM.  2B = 256 rows at B = 128 is one
// workgroup per CU.  Loads are float4 (rows are 16-byte aligned when T % 4 == 0), reductions are wave64 butterflies
// joined through LDS in a fixed order — deterministic, and every thread of the workgroup sees the same scalar, which
// keeps the data-dependent iteration counts workgroup-uniform.
//
// No sort: see include/advstep_fab.h.  Algorithmic bytes per row sample: hyperplane 8 B, projection 12 B (t, w in, d
// out; the re-reads of the fixed-point iteration hit L2), combine 20 B, backward step 8-20 B.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_fab.h"

namespace {

constexpr int kRow = 1024;       // threads per row workgroup
constexpr int kRowWaves = kRow / 64;
constexpr int kMaxNewton = 64;   // the iteration is finite (<= number of breakpoints); in practice 3-8 passes
constexpr float kBig = 1e12f;

inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct Sum {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct MaxNan {  // torch.max semantics: NaN wins
    __device__ __forceinline__ float operator()(float a, float b) const {
        return (a != a) ? a : ((b != b) ? b : fmaxf(a, b));
    }
};

// Reduce NV per-thread values over the workgroup; every thread receives the results.  lds: NV * kRowWaves floats.
template <int NV, class Op>
__device__ __forceinline__ void row_reduce(float (&v)[NV], Op op, float *lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x = op(x, __shfl_xor(x, off, 64));
        if (lane == 0) lds[k * kRowWaves + wave] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float r = lds[k * kRowWaves];
#pragma unroll
        for (int w = 1; w < kRowWaves; ++w) r = op(r, lds[k * kRowWaves + w]);
        v[k] = r;
    }
    __syncthreads();
}

// f(a_i, b_i) over a row pair; VEC: float4 loads.
template <bool VEC, class F>
__device__ __forceinline__ void visit2(const float *__restrict__ a, const float *__restrict__ b, int64_t T, F f) {
    if constexpr (VEC) {
        const float4 *a4 = reinterpret_cast<const float4 *>(a);
        const float4 *b4 = reinterpret_cast<const float4 *>(b);
        const int64_t n4 = T >> 2;
        for (int64_t q = threadIdx.x; q < n4; q += kRow) {
            const float4 x = a4[q], y = b4[q];
            f(x.x, y.x);
            f(x.y, y.y);
            f(x.z, y.z);
            f(x.w, y.w);
        }
    } else {
        for (int64_t i = threadIdx.x; i < T; i += kRow) f(a[i], b[i]);
    }
}

// out_i = f(a_i, b_i), same traversal.
template <bool VEC, class F>
__device__ __forceinline__ void map2(const float *a, const float *b, float *out, int64_t T, F f) {  // out may alias a / b
    if constexpr (VEC) {
        const float4 *a4 = reinterpret_cast<const float4 *>(a);
        const float4 *b4 = reinterpret_cast<const float4 *>(b);
        float4 *o4 = reinterpret_cast<float4 *>(out);
        const int64_t n4 = T >> 2;
        for (int64_t q = threadIdx.x; q < n4; q += kRow) {
            const float4 x = a4[q], y = b4[q];
            float4 o;
            o.x = f(x.x, y.x);
            o.y = f(x.y, y.y);
            o.z = f(x.z, y.z);
            o.w = f(x.w, y.w);
            o4[q] = o;
        }
    } else {
        for (int64_t i = threadIdx.x; i < T; i += kRow) out[i] = f(a[i], b[i]);
    }
}

__device__ __forceinline__ float dual_norm_finish(float v, int kind) { return kind == ADVSTEP_FAB_L2 ? sqrtf(v) : v; }

// ---- hyperplane: row statistics of gz + the 2-logit selection ------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kRow) void fab_hyperplane_kernel(const float *__restrict__ gz, const float *__restrict__ x,
                                                              const float *__restrict__ z,
                                                              const int64_t *__restrict__ labels, float *wscale, float *b,
                                                              float *gnorm, float *gdot, int64_t B, int64_t T, int kind) {
    __shared__ float lds[2 * kRowWaves];
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        const float *g = gz + row * T, *xr = x + row * T;
        float dot[1] = {0.0f}, nrm[1] = {0.0f};
        if (kind == ADVSTEP_FAB_LINF) {
            visit2<VEC>(g, xr, T, [&](float gi, float xi) { dot[0] += gi * xi; nrm[0] += fabsf(gi); });
        } else if (kind == ADVSTEP_FAB_L2) {
            visit2<VEC>(g, xr, T, [&](float gi, float xi) { dot[0] += gi * xi; nrm[0] += gi * gi; });
        } else {
            visit2<VEC>(g, xr, T, [&](float gi, float xi) { dot[0] += gi * xi; nrm[0] = MaxNan()(nrm[0], fabsf(gi)); });
        }
        row_reduce<1>(dot, Sum(), lds);
        if (kind == ADVSTEP_FAB_L1) row_reduce<1>(nrm, MaxNan(), lds + kRowWaves);
        else row_reduce<1>(nrm, Sum(), lds + kRowWaves);
        if (threadIdx.x == 0) {
            const float n = dual_norm_finish(nrm[0], kind);
            if (gnorm) gnorm[row] = n;
            if (gdot) gdot[row] = dot[0];
            if (z && labels) {
                // y = [-z, z]; column k's gradient is (k ? +1 : -1) * gz
                const int la = labels[row] != 0;
                const float zz = z[row];
                const float y0 = -zz, y1 = zz, yla = la ? y1 : y0;
                const float s_la = la ? 1.0f : -1.0f;
                float dist[2], dfv[2], cv[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float c = (k ? 1.0f : -1.0f) - s_la;          // 0 or +-2
                    float df = (k ? y1 : y0) - yla;
                    if (k == la) df = 1e10f;
                    dfv[k] = df;
                    cv[k] = c;
                    dist[k] = fabsf(df) / (1e-12f + fabsf(c) * n);       // |c| in {0, 2}: scaling commutes with the norm
                }
                // torch.min over dim 1: first minimum, NaN wins
                int ind = 0;
                if (!(dist[0] != dist[0]) && ((dist[1] != dist[1]) || dist[1] < dist[0])) ind = 1;
                wscale[row] = cv[ind];
                b[row] = -dfv[ind] + cv[ind] * dot[0];
            }
        }
    }
}

// ---- projections ---------------------------------------------------------------------------------------------------
// Fixed point of  lam = (target - sum_{cap<=lam} wt*cap) / sum_{cap>lam} wt,  started from the all-active guess.
// `pass(lam, A, S, cnt)` streams the row once.  Monotone (lam only grows), ends when the active count repeats.
template <class Pass>
__device__ __forceinline__ float waterfill(float target, float total_weight, float n_active, Pass pass) {
    float lam = target / total_weight;
    float prev = n_active;
    for (int it = 0; it < kMaxNewton; ++it) {
        float A, S, cnt;
        pass(lam, A, S, cnt);
        if (cnt == prev) break;
        if (!(S > 0.0f)) return INFINITY;
        lam = fmaxf(lam, (target - A) / S);
        prev = cnt;
    }
    return lam;
}

template <bool VEC>
__global__ __launch_bounds__(kRow) void fab_projection_linf_kernel(const float *__restrict__ t,
                                                                   const float *__restrict__ w,
                                                                   const float *__restrict__ wscale,
                                                                   const float *__restrict__ b, float *__restrict__ d,
                                                                   float *__restrict__ dnorm, int64_t R, int64_t w_rows,
                                                                   int64_t T) {
    __shared__ float lds[5 * kRowWaves];
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const float *tr = t + row * T, *wr = w + (row % w_rows) * T;
        const float sc = wscale ? wscale[row % w_rows] : 1.0f;
        // pass 1: w.t, sum|w|, and sum|w| * room for both orientations of the hyperplane, count of w != 0
        float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        visit2<VEC>(tr, wr, T, [&](float ti, float wraw) {
            const float wi = sc * wraw, aw = fabsf(wi);
            acc[0] += wi * ti;
            acc[1] += aw;
            acc[2] += aw * (wi < 0.0f ? 1.0f - ti : ti);
            acc[3] += aw * (wi > 0.0f ? 1.0f - ti : ti);
            acc[4] += (wi != 0.0f) ? 1.0f : 0.0f;
        });
        row_reduce<5>(acc, Sum(), lds);
        const float c = acc[0] - b[row];
        const bool keep = c >= 0.0f;                  // fab.py:566: sign = 2 * ((w*t).sum(1) - b >= 0) - 1
        const float beta = fabsf(c);                  // -(b*sign - (w*sign * t).sum(1))
        const float reach = keep ? acc[2] : acc[3];   // -b0: what moving every coordinate to its box face buys
        float lam = INFINITY;
        if (reach - beta > 0.0f) {                    // fab.py:588  b - b0 > 0
            lam = waterfill(beta, acc[1], acc[4], [&](float cur, float &A, float &S, float &cnt) {
                float v[3] = {0.0f, 0.0f, 0.0f};
                visit2<VEC>(tr, wr, T, [&](float ti, float wraw) {
                    const float wi = sc * wraw, aw = fabsf(wi);
                    const bool up = keep ? wi < 0.0f : wi > 0.0f;
                    const float p = up ? 1.0f - ti : ti;
                    const bool active = p > cur;
                    v[0] += active ? 0.0f : aw * p;
                    v[1] += active ? aw : 0.0f;
                    v[2] += (active && wi != 0.0f) ? 1.0f : 0.0f;
                });
                row_reduce<3>(v, Sum(), lds);
                A = v[0];
                S = v[1];
                cnt = v[2];
            });
            lam = fmaxf(lam, 0.0f);                   // clamp_min(lmbd_opt, 0)
        }
        float mx[1] = {0.0f};
        map2<VEC>(tr, wr, d + row * T, T, [&](float ti, float wraw) {
            const float wi = sc * wraw;
            const bool up = keep ? wi < 0.0f : wi > 0.0f;
            const float p = up ? 1.0f - ti : ti;
            const float m = fminf(lam, p);            // lam is never NaN here; p NaN propagates below
            float di = (p != p) ? p : (up ? m : -m);
            if (wi == 0.0f) di = 0.0f;
            mx[0] = MaxNan()(mx[0], fabsf(di));
            return di;
        });
        row_reduce<1>(mx, MaxNan(), lds);
        if (threadIdx.x == 0 && dnorm) dnorm[row] = mx[0];
    }
}

// fab.py:626-628: r = max(t/w, (t-1)/w) clamped to +-1e12, 1e12 where |w| < 1e-8, -1e12 -> 1e12
__device__ __forceinline__ float l2_ratio(float ti, float wi) {
    const float q1 = ti / wi, q2 = (ti - 1.0f) / wi;
    float r = (q1 != q1 || q2 != q2) ? NAN : fmaxf(q1, q2);
    r = (r != r) ? r : fminf(fmaxf(r, -kBig), kBig);
    if (fabsf(wi) < 1e-8f) r = kBig;
    if (r == -kBig) r = kBig;
    return r;
}

template <bool VEC>
__global__ __launch_bounds__(kRow) void fab_projection_l2_kernel(const float *__restrict__ t, const float *__restrict__ w,
                                                                 const float *__restrict__ wscale,
                                                                 const float *__restrict__ b, float *__restrict__ d,
                                                                 float *__restrict__ dnorm, int64_t R, int64_t w_rows,
                                                                 int64_t T) {
    __shared__ float lds[4 * kRowWaves];
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const float *tr = t + row * T, *wr = w + (row % w_rows) * T;
        const float sc = wscale ? wscale[row % w_rows] : 1.0f;
        // pass 1: w.t, sum w^2, and sum (r w) w over the movable coordinates for both orientations
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        visit2<VEC>(tr, wr, T, [&](float ti, float wraw) {
            const float wi = sc * wraw;
            acc[0] += wi * ti;
            acc[1] += wi * wi;
            if (fabsf(wi) > 1e-8f) {
                acc[2] += (l2_ratio(ti, wi) * wi) * wi;
                acc[3] += (l2_ratio(ti, -wi) * wi) * wi;
            }
        });
        row_reduce<4>(acc, Sum(), lds);
        const float c0 = acc[0] - b[row];
        const float sg = c0 >= 0.0f ? 1.0f : -1.0f;
        const float c = fabsf(c0);
        const float reach = sg > 0.0f ? acc[2] : acc[3];
        float alpha = INFINITY;
        if (!(c - reach > 0.0f)) {                   // fab.py:642  c3 = (d*w).sum + c > 0  -> every coordinate to its face
            alpha = waterfill(c, acc[1], (float)T, [&](float cur, float &A, float &S, float &cnt) {
                float v[3] = {0.0f, 0.0f, 0.0f};
                visit2<VEC>(tr, wr, T, [&](float ti, float wraw) {
                    const float wi = sg * (sc * wraw);
                    const float r = l2_ratio(ti, wi), w2 = wi * wi;
                    const bool active = r > cur;
                    v[0] += active ? 0.0f : w2 * r;
                    v[1] += active ? w2 : 0.0f;
                    v[2] += active ? 1.0f : 0.0f;
                });
                row_reduce<3>(v, Sum(), lds);
                A = v[0];
                S = v[1];
                cnt = v[2];
            });
        }
        float ss[1] = {0.0f};
        map2<VEC>(tr, wr, d + row * T, T, [&](float ti, float wraw) {
            const float wi = sg * (sc * wraw);
            const float r = l2_ratio(ti, wi);
            float di = (alpha > r) ? -(r * wi) : -(alpha * wi);   // fab.py:666-667
            if (!(fabsf(wi) > 1e-8f)) di = 0.0f;
            ss[0] += di * di;
            return di;
        });
        row_reduce<1>(ss, Sum(), lds);
        if (threadIdx.x == 0 && dnorm) dnorm[row] = sqrtf(ss[0]);
    }
}

__device__ __forceinline__ float l1_ratio(float wi) {  // fab.py:681  r = (1 / w).abs().clamp_max(1e12)
    return fminf(fabsf(1.0f / wi), kBig);
}
__device__ __forceinline__ float l1_gain(float ti, float wi) { return fminf(-wi * ti, wi * (1.0f - ti)); }
__device__ __forceinline__ float l1_face(float ti, float wi) {
    return (wi != 0.0f) ? ((wi < 0.0f ? 1.0f : 0.0f) - ti) : 0.0f;
}

// Greedy L1: coordinates in order of increasing key r = |1/w| (= decreasing |w|, index order within equal keys) move to
// their face while the residual stays positive.  Keys are non-negative floats, so their bit patterns order like
// integers: the last key whose residual-before is positive is found kDigit bits at a time (a histogram of the gains
// over the 2^kDigit - 1 candidate prefixes per streaming pass, 11 passes for the 31 key bits), then the tie group is
// walked in index order with a workgroup prefix scan.  The output row doubles as scratch for the keys (one division
// per coordinate in total), so d must not alias t or w.
constexpr int kDigit = 3, kBuckets = (1 << kDigit) - 1;

template <bool VEC>
__global__ __launch_bounds__(kRow) void fab_projection_l1_kernel(const float *__restrict__ t, const float *__restrict__ w,
                                                                 const float *__restrict__ wscale,
                                                                 const float *__restrict__ b, float *d,
                                                                 float *__restrict__ dnorm, int64_t R, int64_t w_rows,
                                                                 int64_t T) {
    __shared__ float lds[kBuckets * kRowWaves];
    __shared__ float wave_tot[kRowWaves];
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const float *tr = t + row * T, *wr = w + (row % w_rows) * T;
        float *dr = d + row * T;
        const float sc = wscale ? wscale[row % w_rows] : 1.0f;
        // pass 1: w.t, total gain for either orientation; keys into the output row
        float acc[3] = {0.0f, 0.0f, 0.0f};
        map2<VEC>(tr, wr, dr, T, [&](float ti, float wraw) {
            const float wi = sc * wraw;
            acc[0] += wi * ti;
            acc[1] += l1_gain(ti, wi);
            acc[2] += l1_gain(ti, -wi);
            return l1_ratio(wi);
        });
        row_reduce<3>(acc, Sum(), lds);
        const float c0 = acc[0] - b[row];
        const float sg = c0 >= 0.0f ? 1.0f : -1.0f;
        const float c = fabsf(c0);
        const float total = c + (sg > 0.0f ? acc[1] : acc[2]);   // s[:, -1]
        float sum_abs[1] = {0.0f};
        if (!(total < 0.0f)) {
            // the farthest corner does not reach the hyperplane: every coordinate to its face (fab.py:686)
            map2<VEC>(tr, wr, dr, T, [&](float ti, float wraw) {
                const float wi = sg * (sc * wraw);
                float di = l1_face(ti, wi);
                if (!(fabsf(wi) > 1e-8f)) di = 0.0f;
                sum_abs[0] += fabsf(di);
                return di;
            });
        } else {
            // largest key bits rho with  c + sum_{key < rho} gain > 0;  `before` is that residual
            uint32_t rho = 0;
            float before = c;
            if (c > 0.0f) {
                for (int top = 31; top > 0; top -= kDigit) {          // bits [shift, top) of the key
                    const int shift = top >= kDigit ? top - kDigit : 0;
                    const uint32_t limit = (1u << (top - shift)) - 1u; // candidates j = 1 .. limit
                    float h[kBuckets];
#pragma unroll
                    for (int q = 0; q < kBuckets; ++q) h[q] = 0.0f;
                    auto tally = [&](float ti, float wraw, float ri) {
                        const uint32_t key = __float_as_uint(ri);
                        if (key >= rho) {
                            const uint32_t q = (key - rho) >> shift;   // contributes to every candidate j > q
                            const float gn = l1_gain(ti, sg * (sc * wraw));
#pragma unroll
                            for (int k = 0; k < kBuckets; ++k) h[k] += (q == (uint32_t)k) ? gn : 0.0f;
                        }
                    };
                    if constexpr (VEC) {
                        const int64_t n4 = T >> 2;
                        for (int64_t q4 = threadIdx.x; q4 < n4; q4 += kRow) {
                            const float4 a = reinterpret_cast<const float4 *>(tr)[q4];
                            const float4 e = reinterpret_cast<const float4 *>(wr)[q4];
                            const float4 r = reinterpret_cast<const float4 *>(dr)[q4];
                            tally(a.x, e.x, r.x);
                            tally(a.y, e.y, r.y);
                            tally(a.z, e.z, r.z);
                            tally(a.w, e.w, r.w);
                        }
                    } else {
                        for (int64_t i = threadIdx.x; i < T; i += kRow) tally(tr[i], wr[i], dr[i]);
                    }
                    row_reduce<kBuckets>(h, Sum(), lds);
                    float run = before;
                    uint32_t pick = 0;
                    for (uint32_t j = 1; j <= limit; ++j) {            // residual before candidate j = before + h[0..j)
                        run += h[j - 1];
                        if (run > 0.0f) {
                            pick = j;
                            before = run;
                        } else {
                            break;
                        }
                    }
                    rho |= pick << shift;
                    if (shift == 0) break;
                }
            }
            // every key below rho moves to its face, every key above stays; the tie group at rho holds the coordinate
            // that crosses the hyperplane.  One coalesced pass writes everything but the tie group and counts it.
            float ties[1] = {0.0f};
            float tie_t = 0.0f, tie_w = 0.0f;
            int64_t tie_i = -1;
            auto settle = [&](int64_t i, float ti, float wraw, float ri) {
                const float wi = sg * (sc * wraw);
                const uint32_t key = __float_as_uint(ri);
                float di = 0.0f;
                if (key < rho) di = l1_face(ti, wi);
                if (key == rho) {
                    ties[0] += 1.0f;
                    tie_i = i;
                    tie_t = ti;
                    tie_w = wi;
                }
                if (!(fabsf(wi) > 1e-8f)) di = 0.0f;
                sum_abs[0] += fabsf(di);
                return di;
            };
            if constexpr (VEC) {
                const int64_t n4 = T >> 2;
                for (int64_t q4 = threadIdx.x; q4 < n4; q4 += kRow) {
                    const float4 a = reinterpret_cast<const float4 *>(tr)[q4];
                    const float4 e = reinterpret_cast<const float4 *>(wr)[q4];
                    const float4 r = reinterpret_cast<const float4 *>(dr)[q4];
                    float4 o;
                    o.x = settle(4 * q4, a.x, e.x, r.x);
                    o.y = settle(4 * q4 + 1, a.y, e.y, r.y);
                    o.z = settle(4 * q4 + 2, a.z, e.z, r.z);
                    o.w = settle(4 * q4 + 3, a.w, e.w, r.w);
                    reinterpret_cast<float4 *>(dr)[q4] = o;
                }
            } else {
                for (int64_t i = threadIdx.x; i < T; i += kRow) dr[i] = settle(i, tr[i], wr[i], dr[i]);
            }
            row_reduce<1>(ties, Sum(), lds);
            if (ties[0] == 1.0f) {
                // the usual case: the tie group is one coordinate, the one that would overshoot (fab.py:711,715)
                if (tie_i >= 0) {
                    const float after = before + l1_gain(tie_t, tie_w);
                    float di = after > 0.0f ? l1_face(tie_t, tie_w) : (before > 0.0f ? -before / tie_w : 0.0f);
                    if (!(fabsf(tie_w) > 1e-8f)) di = 0.0f;
                    dr[tie_i] = di;
                    sum_abs[0] += fabsf(di);
                }
            } else if (ties[0] > 1.0f) {
                // equal keys: walk the group in index order (contiguous chunk per thread, exclusive workgroup prefix of
                // the group's gains); keys are recomputed, the output row no longer holds them
                __syncthreads();
                const int64_t chunk = (T + kRow - 1) / kRow;
                const int64_t lo = (int64_t)threadIdx.x * chunk, hi = lo + chunk < T ? lo + chunk : T;
                float local = 0.0f;
                for (int64_t i = lo; i < hi; ++i) {
                    const float wi = sg * (sc * wr[i]);
                    if (__float_as_uint(l1_ratio(wi)) == rho) local += l1_gain(tr[i], wi);
                }
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
                float incl = local;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const float up = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += up;
                }
                if (lane == 63) wave_tot[wave] = incl;
                __syncthreads();
                float run = before;
                for (int wv = 0; wv < wave; ++wv) run += wave_tot[wv];
                run += incl - local;
                for (int64_t i = lo; i < hi; ++i) {
                    const float ti = tr[i], wi = sg * (sc * wr[i]);
                    if (__float_as_uint(l1_ratio(wi)) != rho) continue;
                    const float after = run + l1_gain(ti, wi);
                    float di = after > 0.0f ? l1_face(ti, wi) : (run > 0.0f ? -run / wi : 0.0f);
                    run = after;
                    if (!(fabsf(wi) > 1e-8f)) di = 0.0f;
                    dr[i] = di;
                    sum_abs[0] += fabsf(di);
                }
            }
        }
        row_reduce<1>(sum_abs, Sum(), lds);
        if (threadIdx.x == 0 && dnorm) dnorm[row] = sum_abs[0];
        __syncthreads();
    }
}

// ---- combine -----------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kRow) void fab_combine_kernel(const float *__restrict__ x1, const float *__restrict__ x0,
                                                           const float *__restrict__ d1, const float *__restrict__ d2,
                                                           const float *__restrict__ n1, const float *__restrict__ n2,
                                                           float *out, int64_t B, int64_t T, float eta, float alpha_max) {
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        const float a1 = fmaxf(n1[row], 1e-8f), a2 = fmaxf(n2[row], 1e-8f);
        float alpha = a1 / (a1 + a2);
        alpha = (alpha != alpha) ? alpha : fminf(fmaxf(alpha, 0.0f), alpha_max);
        const float keep = 1.0f - alpha;
        auto f = [&](float p1, float p0, float m1, float m2) {
            const float v = (p1 + eta * m1) * keep + (p0 + m2 * eta) * alpha;
            return (v != v) ? v : fminf(fmaxf(v, 0.0f), 1.0f);
        };
        const float *p1 = x1 + row * T, *p0 = x0 + row * T, *m1 = d1 + row * T, *m2 = d2 + row * T;
        float *o = out + row * T;
        if constexpr (VEC) {
            const int64_t n4 = T >> 2;
            for (int64_t q = threadIdx.x; q < n4; q += kRow) {
                const float4 a = reinterpret_cast<const float4 *>(p1)[q], c = reinterpret_cast<const float4 *>(p0)[q];
                const float4 e = reinterpret_cast<const float4 *>(m1)[q], g = reinterpret_cast<const float4 *>(m2)[q];
                float4 r;
                r.x = f(a.x, c.x, e.x, g.x);
                r.y = f(a.y, c.y, e.y, g.y);
                r.z = f(a.z, c.z, e.z, g.z);
                r.w = f(a.w, c.w, e.w, g.w);
                reinterpret_cast<float4 *>(o)[q] = r;
            }
        } else {
            for (int64_t i = threadIdx.x; i < T; i += kRow) o[i] = f(p1[i], p0[i], m1[i], m2[i]);
        }
    }
}

// ---- backward step -------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kRow) void fab_backward_step_kernel(float *x1, const float *__restrict__ x0, float *adv,
                                                                 float *res2, const uint8_t *__restrict__ is_adv,
                                                                 int64_t B, int64_t T, float beta, int kind) {
    __shared__ float lds[kRowWaves];
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        if (!is_adv[row]) continue;   // workgroup-uniform
        float *p1 = x1 + row * T;
        const float *p0 = x0 + row * T;
        float *pa = adv + row * T;
        float v[1] = {0.0f};
        if (kind == ADVSTEP_FAB_LINF) {
            visit2<VEC>(p1, p0, T, [&](float a, float c) { v[0] = MaxNan()(v[0], fabsf(a - c)); });
            row_reduce<1>(v, MaxNan(), lds);
        } else if (kind == ADVSTEP_FAB_L2) {
            visit2<VEC>(p1, p0, T, [&](float a, float c) { v[0] += (a - c) * (a - c); });
            row_reduce<1>(v, Sum(), lds);
            v[0] = sqrtf(v[0]);
        } else {
            visit2<VEC>(p1, p0, T, [&](float a, float c) { v[0] += fabsf(a - c); });
            row_reduce<1>(v, Sum(), lds);
        }
        const float tn = v[0], best = res2[row];
        const bool better = tn < best, worse = tn >= best;   // both false for a NaN norm: the reference's masks zero adv
        if (!worse) {
            if (better) map2<VEC>(p1, p0, pa, T, [&](float a, float) { return a; });
            else map2<VEC>(p1, pa, pa, T, [&](float a, float o) { return a * 0.0f + o * 0.0f; });
        }
        map2<VEC>(p1, p0, p1, T, [&](float a, float c) { return c + (a - c) * beta; });
        __syncthreads();
        if (threadIdx.x == 0) res2[row] = better ? tn : (worse ? best : tn * 0.0f + best * 0.0f);
    }
}

inline unsigned grid_rows(int64_t rows) { return (unsigned)(rows < 65535 ? rows : 65535); }

}  // namespace

#define FAB_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

int advstep_fab_hyperplane_f32(const float *gz, const float *x, const float *z, const int64_t *labels, float *wscale,
                               float *b, float *gnorm, float *gdot, int64_t B, int64_t T, int norm_kind,
                               advstep_stream_t stream) {
    FAB_REQUIRE(B >= 0 && T >= 0 && norm_kind >= 0 && norm_kind <= 2);
    if (B == 0) return ADVSTEP_OK;
    FAB_REQUIRE(gz && x && ((z == nullptr) == (labels == nullptr)) && (!z || (wscale && b)));
    const bool vec = (T % 4 == 0) && aligned16(gz) && aligned16(x);
    if (vec)
        hipLaunchKernelGGL(fab_hyperplane_kernel<true>, dim3(grid_rows(B)), dim3(kRow), 0, as_stream(stream), gz, x, z, labels,
                           wscale, b, gnorm, gdot, B, T, norm_kind);
    else
        hipLaunchKernelGGL(fab_hyperplane_kernel<false>, dim3(grid_rows(B)), dim3(kRow), 0, as_stream(stream), gz, x, z,
                           labels, wscale, b, gnorm, gdot, B, T, norm_kind);
    return status_after_launch();
}

int advstep_fab_projection_f32(const float *t, const float *w, const float *wscale, const float *b, float *d,
                               float *dnorm, int64_t R, int64_t w_rows, int64_t T, int norm_kind,
                               advstep_stream_t stream) {
    FAB_REQUIRE(R >= 0 && T >= 0 && w_rows >= 0 && norm_kind >= 0 && norm_kind <= 2);
    if (R == 0) return ADVSTEP_OK;
    FAB_REQUIRE(t && w && b && d && d != t && d != w && w_rows >= 1 && T >= 1 && T < (int64_t(1) << 24));
    const bool vec = (T % 4 == 0) && aligned16(t) && aligned16(w) && aligned16(d);
    const dim3 grid(grid_rows(R)), block(kRow);
    hipStream_t st = as_stream(stream);
    if (norm_kind == ADVSTEP_FAB_LINF) {
        if (vec) hipLaunchKernelGGL(fab_projection_linf_kernel<true>, grid, block, 0, st, t, w, wscale, b, d, dnorm, R, w_rows, T);
        else hipLaunchKernelGGL(fab_projection_linf_kernel<false>, grid, block, 0, st, t, w, wscale, b, d, dnorm, R, w_rows, T);
    } else if (norm_kind == ADVSTEP_FAB_L2) {
        if (vec) hipLaunchKernelGGL(fab_projection_l2_kernel<true>, grid, block, 0, st, t, w, wscale, b, d, dnorm, R, w_rows, T);
        else hipLaunchKernelGGL(fab_projection_l2_kernel<false>, grid, block, 0, st, t, w, wscale, b, d, dnorm, R, w_rows, T);
    } else {
        if (vec) hipLaunchKernelGGL(fab_projection_l1_kernel<true>, grid, block, 0, st, t, w, wscale, b, d, dnorm, R, w_rows, T);
        else hipLaunchKernelGGL(fab_projection_l1_kernel<false>, grid, block, 0, st, t, w, wscale, b, d, dnorm, R, w_rows, T);
    }
    return status_after_launch();
}

int advstep_fab_combine_f32(const float *x1, const float *x0, const float *d1, const float *d2, const float *n1,
                            const float *n2, float *out, int64_t B, int64_t T, float eta, float alpha_max,
                            advstep_stream_t stream) {
    FAB_REQUIRE(B >= 0 && T >= 0);
    if (B == 0 || T == 0) return ADVSTEP_OK;
    FAB_REQUIRE(x1 && x0 && d1 && d2 && n1 && n2 && out);
    const bool vec = (T % 4 == 0) && aligned16(x1) && aligned16(x0) && aligned16(d1) && aligned16(d2) && aligned16(out);
    if (vec)
        hipLaunchKernelGGL(fab_combine_kernel<true>, dim3(grid_rows(B)), dim3(kRow), 0, as_stream(stream), x1, x0, d1, d2, n1,
                           n2, out, B, T, eta, alpha_max);
    else
        hipLaunchKernelGGL(fab_combine_kernel<false>, dim3(grid_rows(B)), dim3(kRow), 0, as_stream(stream), x1, x0, d1, d2, n1,
                           n2, out, B, T, eta, alpha_max);
    return status_after_launch();
}

int advstep_fab_backward_step_f32(float *x1, const float *x0, float *adv, float *res2, const uint8_t *is_adv, int64_t B,
                                  int64_t T, float beta, int norm_kind, advstep_stream_t stream) {
    FAB_REQUIRE(B >= 0 && T >= 0 && norm_kind >= 0 && norm_kind <= 2);
    if (B == 0) return ADVSTEP_OK;
    FAB_REQUIRE(x1 && x0 && adv && res2 && is_adv);
    const bool vec = (T % 4 == 0) && aligned16(x1) && aligned16(x0) && aligned16(adv);
    if (vec)
        hipLaunchKernelGGL(fab_backward_step_kernel<true>, dim3(grid_rows(B)), dim3(kRow), 0, as_stream(stream), x1, x0, adv,
                           res2, is_adv, B, T, beta, norm_kind);
    else
        hipLaunchKernelGGL(fab_backward_step_kernel<false>, dim3(grid_rows(B)), dim3(kRow), 0, as_stream(stream), x1, x0, adv,
                           res2, is_adv, B, T, beta, norm_kind);
    return status_after_launch();
}

}  // extern "C"
