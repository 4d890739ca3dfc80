// lcnn_lstm.hip — recurrent part of LCNN's bidirectional LSTM layers on gfx950, forward and input-backward
// (C ABI: include/advstep_lcnn.h; reference: src/models/lcnn.py:24-46 `BLSTMLayer` = torch.nn.LSTM(160, 80,
// bidirectional=True) over 25 frames).
//
// Under PyTorch-ROCm the layer runs through MIOpen's RNN path: ~100 tiny GEMM + gate kernels per direction pair and
// pass (8 000+ launches of 3-5 us per 40-iteration PGD step, ~20 % of the step once the convolution side is fused).
// The recurrence of one utterance is independent of every other utterance, so here ONE workgroup owns one
// (utterance, direction) for all T steps: 4H = 320 threads, thread j keeps row j of W_hh (H = 80 floats) in
// registers, h_{t-1} lives in LDS (broadcast reads), two LDS-only barriers per step, each thread applies its own gate's
// nonlinearity; the backward reduces the saved activations to coefficients one step ahead of the recurrence.  256 workgroups at B = 128: one per CU.
// The input projection W_ih x + b (all steps, both directions) stays ONE rocBLAS GEMM on the torch side, as does its
// backward.  Gate order and formulas are torch.nn.LSTM's (i, f, g, o).
// Latency-bound by design (25 dependent steps); neither HBM nor MFMA is the limiter at this size.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "advstep_lcnn.h"

namespace {

inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access
// (s_waitcnt vmcnt(0)): inside the step loops that put each step's stores — and the prefetch of the next step's inputs —
// on the critical path of a 25-step recurrence.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// gx (T, B, D, 4H), w_hh (D, 4H, H), out (T, B, D*H), gates (T, B, D, 4H), cell (T, B, D, H)
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_forward_kernel(const float *__restrict__ gx,
                                                             const float *__restrict__ w_hh, float *__restrict__ out,
                                                             float *__restrict__ gates, float *__restrict__ cell,
                                                             int T, int B, int D) {
    __shared__ __attribute__((aligned(16))) float h_s[H];
    __shared__ float pre[4 * H];
    const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
    float w[H];
    {
        const float *wr = w_hh + ((int64_t)d * 4 * H + j) * H;
#pragma unroll
        for (int k = 0; k < H; ++k) w[k] = wr[k];
    }
    if (j < H) h_s[j] = 0.0f;
    float c = 0.0f;
    const bool is_g = j >= 2 * H && j < 3 * H;   // gate order i, f, g, o
    __syncthreads();
    float gx_next = T > 0 ? gx[(((int64_t)(d == 0 ? 0 : T - 1) * B + b) * D + d) * 4 * H + j] : 0.0f;
    for (int step = 0; step < T; ++step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int64_t row = ((int64_t)t * B + b) * D + d;
        float acc = gx_next;
        if (step + 1 < T) {  // prefetch the next step's projection while this step computes
            const int tn = d == 0 ? step + 1 : T - 2 - step;
            gx_next = gx[(((int64_t)tn * B + b) * D + d) * 4 * H + j];
        }
        // four partial sums: a single 80-long dependent fma chain is latency-, not throughput-bound
        float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4 *>(&h_s[k]);
            acc = fmaf(w[k], hv.x, acc);
            a1 = fmaf(w[k + 1], hv.y, a1);
            a2 = fmaf(w[k + 2], hv.z, a2);
            a3 = fmaf(w[k + 3], hv.w, a3);
        }
        // every thread applies its own gate's nonlinearity (the transcendental work of a step spread over all 320
        // threads instead of 80): sigmoid for i, f, o; tanh(x) = 2 sigmoid(2x) - 1 for g
        const float z = (acc + a1) + (a2 + a3);
        const float sg = 1.0f / (1.0f + expf(is_g ? -2.0f * z : -z));
        pre[j] = is_g ? 2.0f * sg - 1.0f : sg;
        lds_barrier();
        if (j < H) {
            const float ig = pre[j], fg = pre[H + j], gg = pre[2 * H + j], og = pre[3 * H + j];
            c = fg * c + ig * gg;
            const float h = og * (2.0f / (1.0f + expf(-2.0f * c)) - 1.0f);   // og * tanh(c)
            float *gr = gates + row * 4 * H;
            gr[j] = ig;
            gr[H + j] = fg;
            gr[2 * H + j] = gg;
            gr[3 * H + j] = og;
            cell[row * H + j] = c;
            out[((int64_t)t * B + b) * D * H + d * H + j] = h;
            h_s[j] = h;
        }
        lds_barrier();
    }
}

typedef float f32x2_l __attribute__((ext_vector_type(2)));

// dout (T, B, D*H) -> dgx (T, B, D, 4H): gradient w.r.t. the gate pre-activations (what the projection GEMM consumes)
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_backward_kernel(const float *__restrict__ dout,
                                                              const float *__restrict__ w_hh,
                                                              const float *__restrict__ gates,
                                                              const float *__restrict__ cell, float *__restrict__ dgx,
                                                              int T, int B, int D, int64_t dout_t_stride, int64_t dout_b_stride,
                                                              const float *__restrict__ dout_scale) {
    __shared__ __attribute__((aligned(16))) float dg_s[4 * H];
    __shared__ float part[4 * H];
    const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const int k = tid % H, q = tid / H;
    // thread (k, q) holds W_hh[q*H + jj][k], jj < H: its share of the transposed product dh[k] = sum_j dg[j] W_hh[j][k]
    float w[H];
    {
        const float *wc = w_hh + ((int64_t)d * 4 * H + (int64_t)q * H) * H + k;
#pragma unroll
        for (int jj = 0; jj < H; ++jj) w[jj] = wc[(int64_t)jj * H];
    }
    float dc_next = 0.0f;
    // a step's saved activations and incoming gradient (threads < H): loaded one step ahead
    // Per step, threads < H need the saved gates / cell states and dout.  None of that depends on the recurrence
    // (dh, dc), so it is loaded AND reduced to five coefficients one step ahead; what stays on the critical path is
    //     dh = dout + sum of the four partial products;  d_o = dh a_o;  dc = dh a_c + dc_next;  d_{i,f,g} = dc a_{i,f,g}.
    struct Raw {
        float ig, fg, gg, og, c, c_prev, dout;
    };
    struct Coef {
        float a_o, a_c, a_i, a_f, a_g, fg, dout;
    };
    auto load = [&](int step) {                  // issue only: nothing here waits for the values
        Raw v = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (tid < H && step >= 0) {
            const int t = d == 0 ? step : T - 1 - step, j = tid;
            const int64_t row = ((int64_t)t * B + b) * D + d;
            const float *gr = gates + row * 4 * H;
            v.ig = gr[j], v.fg = gr[H + j], v.gg = gr[2 * H + j], v.og = gr[3 * H + j];
            v.c = cell[row * H + j];
            if (step > 0) {
                const int tp = d == 0 ? step - 1 : T - step;
                v.c_prev = cell[(((int64_t)tp * B + b) * D + d) * H + j];
            }
            // strides 0: the same row for every frame / utterance; dout_scale: one factor per utterance (the outer product
            // dz[b] * row[f] of the mean's gradient, formed here instead of by an elementwise launch in front of this kernel)
            v.dout = dout[(int64_t)t * dout_t_stride + (int64_t)b * dout_b_stride + d * H + j];
            if (dout_scale) v.dout = dout_scale[b] * v.dout;
        }
        return v;
    };
    auto reduce = [&](const Raw &r) {
        const float tc = 2.0f / (1.0f + expf(-2.0f * r.c)) - 1.0f;   // tanh(c)
        Coef v;
        v.a_o = tc * r.og * (1.0f - r.og);
        v.a_c = r.og * (1.0f - tc * tc);
        v.a_i = r.gg * r.ig * (1.0f - r.ig);
        v.a_f = r.c_prev * r.fg * (1.0f - r.fg);
        v.a_g = r.ig * (1.0f - r.gg * r.gg);
        v.fg = r.fg;
        v.dout = r.dout;
        return v;
    };
    part[tid] = 0.0f;                            // dh of the step after the last one
    Coef cur = reduce(load(T - 1));
    Raw raw = load(T - 2);                       // three-stage pipeline: loading (n - 2) | reducing (n - 1) | recurrence (n)
    __syncthreads();
    for (int step = T - 1; step >= 0; --step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int64_t row = ((int64_t)t * B + b) * D + d;
        if (tid < H) {
            const int j = tid;
            const float dh = cur.dout + ((part[j] + part[H + j]) + (part[2 * H + j] + part[3 * H + j]));
            const float d_o = dh * cur.a_o;
            const float dc = dh * cur.a_c + dc_next;
            const float d_i = dc * cur.a_i, d_f = dc * cur.a_f, d_g = dc * cur.a_g;
            dc_next = dc * cur.fg;
            dg_s[j] = d_i;
            dg_s[H + j] = d_f;
            dg_s[2 * H + j] = d_g;
            dg_s[3 * H + j] = d_o;
            float *dr = dgx + row * 4 * H;
            dr[j] = d_i;
            dr[H + j] = d_f;
            dr[2 * H + j] = d_g;
            dr[3 * H + j] = d_o;
        }
        lds_barrier();                           // dg_s complete; every read of part[] above is done
        const Raw raw_next = load(step - 2);     // in flight for a whole step before anything reads it
        // all 20 reads of dg_s requested before the first product (left to the compiler they are issued three ahead of their use
        // and the step pays their LDS latencies one after another), products on v_pk_fma_f32: the same four chains, two per
        // instruction — bit-identical sums; 31.7 -> 27.8 us at T = 25, B = 128 (round 3).  The forward kernel restructured the
        // same way, with a unit's four gates in one lane quad (DPP exchange, one barrier per step), measured 0.79 instead of
        // 0.82 us per step but 1 us more up front: 29.6 vs 28.7 us at T = 25 — not kept
        float4 gv[H / 4];
#pragma unroll
        for (int jj = 0; jj < H / 4; ++jj) gv[jj] = *reinterpret_cast<const float4 *>(&dg_s[q * H + 4 * jj]);
        __builtin_amdgcn_sched_barrier(0);
        f32x2_l s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
        for (int jj = 0; jj < H / 4; ++jj) {
            s01 = __builtin_elementwise_fma((f32x2_l){gv[jj].x, gv[jj].y}, (f32x2_l){w[4 * jj], w[4 * jj + 1]}, s01);
            s23 = __builtin_elementwise_fma((f32x2_l){gv[jj].z, gv[jj].w}, (f32x2_l){w[4 * jj + 2], w[4 * jj + 3]}, s23);
        }
        part[tid] = (s01.x + s01.y) + (s23.x + s23.y);
        const Coef nxt = reduce(raw);            // values loaded during the previous step
        lds_barrier();                           // part[] complete; dg_s free for the next step
        cur = nxt;
        raw = raw_next;
    }
}


// ---- what surrounds the two BLSTM layers in LCNN (src/models/lcnn.py:196-205), as three passes instead of ~15 ATen launches -------
//   hidden = conv_out.permute(0, 2, 1, 3).contiguous().view(B, T, C W);  lstm = blstm2(blstm1(hidden));
//   z = Linear((lstm + hidden).mean(1))
// The recurrent kernels work sequence-first, so `hidden` is packed straight into (T, B, C W); the skip connection, the mean over
// frames and the one-row Linear are one pass over the two (T, B, F) tensors; on the way back the mean's gradient is the same
// (B, F) row for every frame (lstm_backward_kernel reads it with a zero frame stride) and the two gradients of `hidden` are
// summed while they are put back into the convolution's (B, C, T, W) layout.

// x4 (B, C, T, W) -> xt (T, B, C W)
__global__ __launch_bounds__(256) void lcnn_tail_pack_kernel(const float *__restrict__ x4, float *__restrict__ xt, int B, int C,
                                                             int T, int W) {
    const int64_t total = (int64_t)T * B * C * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), c = (int)((i / W) % C), b = (int)((i / ((int64_t)W * C)) % B), t = (int)(i / ((int64_t)W * C * B));
        xt[i] = x4[(((int64_t)b * C + c) * T + t) * W + w];
    }
}

// z[b] = bias + sum_k w[k] * (1 / T) sum_t (a[t][b][k] + xt[t][b][k]);  one workgroup per utterance, F <= 256
__global__ __launch_bounds__(256) void lcnn_tail_forward_kernel(const float *__restrict__ a, const float *__restrict__ xt,
                                                                const float *__restrict__ w, const float *__restrict__ bias,
                                                                float *__restrict__ z, int T, int B, int F) {
    __shared__ float red[4];
    const int b = blockIdx.x, k = threadIdx.x;
    float acc = 0.0f;
    if (k < F) {
        // the same sum in the same order, the loads of 8 frames in flight at a time (a plain loop is one L2 round trip per frame:
        // 25 in a row were most of this kernel's 9 us)
        float s = 0.0f;
        for (int t0 = 0; t0 < T; t0 += 8) {
            float av[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u < T ? t0 + u : T - 1;
                const int64_t i = ((int64_t)t * B + b) * F + k;
                av[u] = a[i];
                xv[u] = xt[i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t0 + u < T) s += av[u] + xv[u];
        }
        acc = (s / (float)T) * w[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) z[b] = ((red[0] + red[1]) + (red[2] + red[3])) + (bias ? bias[0] : 0.0f);
}

// dx4 (B, C, T, W) = dxt (T, B, C W) + g0 (B, C W);  dz != null: g0[b][k] = dz[b] * g0[k] (g0 is then ONE row)
__global__ __launch_bounds__(256) void lcnn_tail_unpack_add_kernel(const float *__restrict__ dxt, const float *__restrict__ g0,
                                                                   const float *__restrict__ dz, float *__restrict__ dx4, int B,
                                                                   int C, int T, int W) {
    const int64_t total = (int64_t)B * C * T * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), t = (int)((i / W) % T), c = (int)((i / ((int64_t)W * T)) % C), b = (int)(i / ((int64_t)W * T * C));
        const int k = c * W + w;
        const float g = dz ? dz[b] * g0[k] : g0[(int64_t)b * C * W + k];
        dx4[i] = dxt[((int64_t)t * B + b) * C * W + k] + g;
    }
}

}  // namespace

#define LSTM_REQUIRE(cond) \
    do {                   \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

int advstep_lstm_supported(int64_t H) { return H == 80; }

int advstep_lstm_forward_f32(const float *gx, const float *w_hh, float *out, float *gates, float *cell, int64_t T,
                             int64_t B, int64_t D, int64_t H, advstep_stream_t stream) {
    LSTM_REQUIRE(T >= 0 && B >= 0 && (D == 1 || D == 2) && advstep_lstm_supported(H));
    if (T == 0 || B == 0) return ADVSTEP_OK;
    LSTM_REQUIRE(gx && w_hh && out && gates && cell && T <= INT32_MAX && B <= INT32_MAX);
    hipLaunchKernelGGL(lstm_forward_kernel<80>, dim3((unsigned)B, (unsigned)D), dim3(320), 0, as_stream(stream), gx, w_hh,
                       out, gates, cell, (int)T, (int)B, (int)D);
    return status_after_launch();
}

int advstep_lstm_backward_f32(const float *dout, const float *w_hh, const float *gates, const float *cell, float *dgx,
                              int64_t T, int64_t B, int64_t D, int64_t H, advstep_stream_t stream) {
    LSTM_REQUIRE(T >= 0 && B >= 0 && (D == 1 || D == 2) && advstep_lstm_supported(H));
    if (T == 0 || B == 0) return ADVSTEP_OK;
    LSTM_REQUIRE(dout && w_hh && gates && cell && dgx && T <= INT32_MAX && B <= INT32_MAX);
    hipLaunchKernelGGL(lstm_backward_kernel<80>, dim3((unsigned)B, (unsigned)D), dim3(320), 0, as_stream(stream), dout,
                       w_hh, gates, cell, dgx, (int)T, (int)B, (int)D, (int64_t)B * D * H, (int64_t)D * H,
                       static_cast<const float *>(nullptr));
    return status_after_launch();
}

int advstep_lstm_backward_bcast_f32(const float *dout_row, const float *w_hh, const float *gates, const float *cell,
                                    float *dgx, int64_t T, int64_t B, int64_t D, int64_t H, advstep_stream_t stream) {
    LSTM_REQUIRE(T >= 0 && B >= 0 && (D == 1 || D == 2) && advstep_lstm_supported(H));
    if (T == 0 || B == 0) return ADVSTEP_OK;
    LSTM_REQUIRE(dout_row && w_hh && gates && cell && dgx && T <= INT32_MAX && B <= INT32_MAX);
    hipLaunchKernelGGL(lstm_backward_kernel<80>, dim3((unsigned)B, (unsigned)D), dim3(320), 0, as_stream(stream), dout_row,
                       w_hh, gates, cell, dgx, (int)T, (int)B, (int)D, (int64_t)0, (int64_t)D * H, static_cast<const float *>(nullptr));
    return status_after_launch();
}

int advstep_lstm_backward_outer_f32(const float *dz, const float *row, const float *w_hh, const float *gates, const float *cell,
                                    float *dgx, int64_t T, int64_t B, int64_t D, int64_t H, advstep_stream_t stream) {
    LSTM_REQUIRE(T >= 0 && B >= 0 && (D == 1 || D == 2) && advstep_lstm_supported(H));
    if (T == 0 || B == 0) return ADVSTEP_OK;
    LSTM_REQUIRE(dz && row && w_hh && gates && cell && dgx && T <= INT32_MAX && B <= INT32_MAX);
    hipLaunchKernelGGL(lstm_backward_kernel<80>, dim3((unsigned)B, (unsigned)D), dim3(320), 0, as_stream(stream), row, w_hh, gates,
                       cell, dgx, (int)T, (int)B, (int)D, (int64_t)0, (int64_t)0, dz);
    return status_after_launch();
}

int advstep_lcnn_tail_pack_f32(const float *x4, float *xt, int64_t B, int64_t C, int64_t T, int64_t W, advstep_stream_t stream) {
    LSTM_REQUIRE(B >= 0 && C >= 0 && T >= 0 && W >= 0);
    const int64_t total = B * C * T * W;
    if (total == 0) return ADVSTEP_OK;
    LSTM_REQUIRE(x4 && xt && B <= INT32_MAX && C <= INT32_MAX && T <= INT32_MAX && W <= INT32_MAX);
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(lcnn_tail_pack_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), x4,
                       xt, (int)B, (int)C, (int)T, (int)W);
    return status_after_launch();
}

int advstep_lcnn_tail_forward_f32(const float *a, const float *xt, const float *w, const float *bias, float *z, int64_t T,
                                  int64_t B, int64_t F, advstep_stream_t stream) {
    LSTM_REQUIRE(T >= 1 && B >= 0 && F >= 1 && F <= 256);
    if (B == 0) return ADVSTEP_OK;
    LSTM_REQUIRE(a && xt && w && z && T <= INT32_MAX && B <= INT32_MAX);
    hipLaunchKernelGGL(lcnn_tail_forward_kernel, dim3((unsigned)B), dim3(256), 0, as_stream(stream), a, xt, w, bias, z, (int)T,
                       (int)B, (int)F);
    return status_after_launch();
}

int advstep_lcnn_tail_unpack_add_f32(const float *dxt, const float *g0, float *dx4, int64_t B, int64_t C, int64_t T, int64_t W,
                                     advstep_stream_t stream) {
    LSTM_REQUIRE(B >= 0 && C >= 0 && T >= 0 && W >= 0);
    const int64_t total = B * C * T * W;
    if (total == 0) return ADVSTEP_OK;
    LSTM_REQUIRE(dxt && g0 && dx4 && B <= INT32_MAX && C <= INT32_MAX && T <= INT32_MAX && W <= INT32_MAX);
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(lcnn_tail_unpack_add_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       as_stream(stream), dxt, g0, static_cast<const float *>(nullptr), dx4, (int)B, (int)C, (int)T, (int)W);
    return status_after_launch();
}

int advstep_lcnn_tail_unpack_add_outer_f32(const float *dxt, const float *dz, const float *row, float *dx4, int64_t B, int64_t C,
                                           int64_t T, int64_t W, advstep_stream_t stream) {
    LSTM_REQUIRE(B >= 0 && C >= 0 && T >= 0 && W >= 0);
    const int64_t total = B * C * T * W;
    if (total == 0) return ADVSTEP_OK;
    LSTM_REQUIRE(dxt && dz && row && dx4 && B <= INT32_MAX && C <= INT32_MAX && T <= INT32_MAX && W <= INT32_MAX);
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(lcnn_tail_unpack_add_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       as_stream(stream), dxt, row, dz, dx4, (int)B, (int)C, (int)T, (int)W);
    return status_after_launch();
}

}  // extern "C"
