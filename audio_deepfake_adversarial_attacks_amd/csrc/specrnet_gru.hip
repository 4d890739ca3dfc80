// specrnet_gru.hip — recurrent part of SpecRNet's bidirectional GRU layers on gfx950, forward and input-backward
// (C ABI: include/advstep_lcnn.h; reference: src/models/specrnet.py:121-127,176-177 — nn.GRU(64, 64, num_layers=2,
// batch_first=True, bidirectional=True)).
//
// Under PyTorch-ROCm the two layers run through MIOpen's RNN path: ~400 kernels of ~4 us per forward + backward
// (2.6 ms of the 9.2 ms PGDL2 iteration of BASELINE configs[2] at B = 128).  Same design as lcnn_lstm.hip: ONE workgroup
// owns one (utterance, direction) for all T steps; 3H threads, thread j keeps row j of W_hh in registers, h_{t-1} lives in
// LDS, two LDS-only barriers per step.  The input projections W_ih x + b_ih stay one GEMM on the torch side.
// Gate order and formulas are torch.nn.GRU's (r, z, n):
//     r = sigmoid(gx_r + W_hr h + b_hr)      z = sigmoid(gx_z + W_hz h + b_hz)
//     n = tanh(gx_n + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h
// Kept for the backward: r, z, n and a_n = W_hn h + b_hn per step (4H floats); h_{t-1} is the previous output row.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_lcnn.h"

namespace {

inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

// workgroup barrier that orders LDS traffic only (see lcnn_lstm.hip)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// gx (T, B, D, 3H), w_hh (D, 3H, H), b_hh (D, 3H), out (T, B, D*H), saved (T, B, D, 4H) = r, z, n, a_n
template <int H>
__global__ __launch_bounds__(3 * H) void gru_forward_kernel(const float *__restrict__ gx, const float *__restrict__ w_hh,
                                                            const float *__restrict__ b_hh, float *__restrict__ out,
                                                            float *__restrict__ saved, int T, int B, int D) {
    __shared__ __attribute__((aligned(16))) float h_s[H];
    __shared__ float a_s[3 * H];
    const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
    float w[H];
    {
        const float *wr = w_hh + ((int64_t)d * 3 * H + j) * H;
#pragma unroll
        for (int k = 0; k < H; ++k) w[k] = wr[k];
    }
    const float bj = b_hh[d * 3 * H + j];
    if (j < H) h_s[j] = 0.0f;
    float h = 0.0f;
    __syncthreads();
    auto gx_at = [&](int step) {
        const int t = d == 0 ? step : T - 1 - step;
        return gx[(((int64_t)t * B + b) * D + d) * 3 * H + j];
    };
    float gx_next = T > 0 ? gx_at(0) : 0.0f;
    for (int step = 0; step < T; ++step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int64_t row = ((int64_t)t * B + b) * D + d;
        const float g = gx_next;
        if (step + 1 < T) gx_next = gx_at(step + 1);   // prefetch behind this step's products
        float a0 = bj, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4 *>(&h_s[k]);
            a0 = fmaf(w[k], hv.x, a0);
            a1 = fmaf(w[k + 1], hv.y, a1);
            a2 = fmaf(w[k + 2], hv.z, a2);
            a3 = fmaf(w[k + 3], hv.w, a3);
        }
        const float a = (a0 + a1) + (a2 + a3);
        // r and z threads finish their gate here (one exp each, in parallel); n threads publish a_n
        a_s[j] = j < 2 * H ? 1.0f / (1.0f + expf(-(g + a))) : a;
        lds_barrier();
        if (j >= 2 * H) {                      // the n threads own the state update of unit u = j - 2H
            const int u = j - 2 * H;
            const float r = a_s[u], z = a_s[H + u];
            const float n = 2.0f / (1.0f + expf(-2.0f * (g + r * a))) - 1.0f;   // tanh
            h = (1.0f - z) * n + z * h;
            float *sr = saved + row * 4 * H;
            sr[u] = r;
            sr[H + u] = z;
            sr[2 * H + u] = n;
            sr[3 * H + u] = a;
            out[((int64_t)t * B + b) * D * H + d * H + u] = h;
            h_s[u] = h;
        }
        lds_barrier();
    }
}

// dout (T, B, D*H) -> dgx (T, B, D, 3H): gradient w.r.t. the input projections gx
template <int H>
__global__ __launch_bounds__(3 * H) void gru_backward_kernel(const float *__restrict__ dout,
                                                             const float *__restrict__ w_hh,
                                                             const float *__restrict__ saved,
                                                             const float *__restrict__ out, float *__restrict__ dgx,
                                                             int T, int B, int D) {
    __shared__ __attribute__((aligned(16))) float da_s[3 * H];   // gradient w.r.t. a = W_hh h + b_hh  (r, z, n rows)
    __shared__ float part[3 * H];
    const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const int k = tid % H, q = tid / H;
    // thread (k, q) holds W_hh[q*H + jj][k], jj < H: its share of dh[k] = sum_j da[j] W_hh[j][k]
    float w[H];
    {
        const float *wc = w_hh + ((int64_t)d * 3 * H + (int64_t)q * H) * H + k;
#pragma unroll
        for (int jj = 0; jj < H; ++jj) w[jj] = wc[(int64_t)jj * H];
    }
    struct Raw {
        float r, z, n, a_n, h_prev, dout;
    };
    auto load = [&](int step) {
        Raw v = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (tid < H && step >= 0) {
            const int t = d == 0 ? step : T - 1 - step, u = tid;
            const int64_t row = ((int64_t)t * B + b) * D + d;
            const float *sr = saved + row * 4 * H;
            v.r = sr[u], v.z = sr[H + u], v.n = sr[2 * H + u], v.a_n = sr[3 * H + u];
            if (step > 0) {
                const int tp = d == 0 ? step - 1 : T - step;
                v.h_prev = out[((int64_t)tp * B + b) * D * H + d * H + u];
            }
            v.dout = dout[((int64_t)t * B + b) * D * H + d * H + u];
        }
        return v;
    };
    part[tid] = 0.0f;
    float dh_carry = 0.0f;                       // the z * dh' path into h_{t-1}
    Raw cur = load(T - 1);
    __syncthreads();
    for (int step = T - 1; step >= 0; --step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int64_t row = ((int64_t)t * B + b) * D + d;
        const Raw nxt = load(step - 1);          // in flight while this step computes
        if (tid < H) {
            const int u = tid;
            const float dh = cur.dout + dh_carry + ((part[u] + part[H + u]) + part[2 * H + u]);
            const float dn_pre = dh * (1.0f - cur.z) * (1.0f - cur.n * cur.n);
            const float dz_pre = dh * (cur.h_prev - cur.n) * cur.z * (1.0f - cur.z);
            const float dr_pre = dn_pre * cur.a_n * cur.r * (1.0f - cur.r);
            dh_carry = dh * cur.z;
            da_s[u] = dr_pre;
            da_s[H + u] = dz_pre;
            da_s[2 * H + u] = dn_pre * cur.r;    // through r * a_n
            float *dr = dgx + row * 3 * H;
            dr[u] = dr_pre;
            dr[H + u] = dz_pre;
            dr[2 * H + u] = dn_pre;
        }
        lds_barrier();
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (int jj = 0; jj < H; jj += 4) {
            const float4 gv = *reinterpret_cast<const float4 *>(&da_s[q * H + jj]);
            s0 = fmaf(gv.x, w[jj], s0);
            s1 = fmaf(gv.y, w[jj + 1], s1);
            s2 = fmaf(gv.z, w[jj + 2], s2);
            s3 = fmaf(gv.w, w[jj + 3], s3);
        }
        part[tid] = (s0 + s1) + (s2 + s3);
        lds_barrier();
        cur = nxt;
    }
}

}  // namespace

#define GRU_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return ADVSTEP_EINVAL; \
    } while (0)

extern "C" {

int advstep_gru_supported(int64_t H) { return H == 64; }

int advstep_gru_forward_f32(const float *gx, const float *w_hh, const float *b_hh, float *out, float *saved, int64_t T,
                            int64_t B, int64_t D, int64_t H, advstep_stream_t stream) {
    GRU_REQUIRE(T >= 0 && B >= 0 && (D == 1 || D == 2) && advstep_gru_supported(H));
    if (T == 0 || B == 0) return ADVSTEP_OK;
    GRU_REQUIRE(gx && w_hh && b_hh && out && saved && T <= INT32_MAX && B <= 65535);
    hipLaunchKernelGGL(gru_forward_kernel<64>, dim3((unsigned)B, (unsigned)D), dim3(192), 0, as_stream(stream), gx, w_hh,
                       b_hh, out, saved, (int)T, (int)B, (int)D);
    return status_after_launch();
}

int advstep_gru_backward_f32(const float *dout, const float *w_hh, const float *saved, const float *out, float *dgx,
                             int64_t T, int64_t B, int64_t D, int64_t H, advstep_stream_t stream) {
    GRU_REQUIRE(T >= 0 && B >= 0 && (D == 1 || D == 2) && advstep_gru_supported(H));
    if (T == 0 || B == 0) return ADVSTEP_OK;
    GRU_REQUIRE(dout && w_hh && saved && out && dgx && T <= INT32_MAX && B <= 65535);
    hipLaunchKernelGGL(gru_backward_kernel<64>, dim3((unsigned)B, (unsigned)D), dim3(192), 0, as_stream(stream), dout,
                       w_hh, saved, out, dgx, (int)T, (int)B, (int)D);
    return status_after_launch();
}

}  // extern "C"
