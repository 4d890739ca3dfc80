// lcnn_lstm.hip — recurrent part of LCNN's bidirectional LSTM layers on gfx950, forward and input-backward
// (C ABI: include/advstep_lcnn.h; reference: src/models/lcnn.py:24-46 `BLSTMLayer` = torch.nn.LSTM(160, 80,
// bidirectional=True) over 25 frames).
//
// Under PyTorch-ROCm the layer runs through MIOpen's RNN path: ~100 tiny GEMM + gate kernels per direction pair and
// pass (8 000+ launches of 3-5 us per 40-iteration PGD step, ~20 % of the step once the convolution side is fused).
// The recurrence of one utterance is independent of every other utterance, so here ONE workgroup owns one
// (utterance, direction) for all T steps: 4H = 320 threads, thread j keeps row j of W_hh (H = 80 floats) in
// registers, h_{t-1} lives in LDS (broadcast reads), two barriers per step.  256 workgroups at B = 128: one per CU.
// The input projection W_ih x + b (all steps, both directions) stays ONE rocBLAS GEMM on the torch side, as does its
// backward.  Gate order and formulas are torch.nn.LSTM's (i, f, g, o).
// Latency-bound by design (25 dependent steps); neither HBM nor MFMA is the limiter at this size.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "advstep_lcnn.h"

namespace {

inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

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
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4 *>(&h_s[k]);
            acc = fmaf(w[k], hv.x, acc);
            acc = fmaf(w[k + 1], hv.y, acc);
            acc = fmaf(w[k + 2], hv.z, acc);
            acc = fmaf(w[k + 3], hv.w, acc);
        }
        pre[j] = acc;
        __syncthreads();
        if (j < H) {
            const float ig = sigmoidf_(pre[j]), fg = sigmoidf_(pre[H + j]), gg = tanhf(pre[2 * H + j]),
                        og = sigmoidf_(pre[3 * H + j]);
            c = fg * c + ig * gg;
            const float h = og * tanhf(c);
            float *gr = gates + row * 4 * H;
            gr[j] = ig;
            gr[H + j] = fg;
            gr[2 * H + j] = gg;
            gr[3 * H + j] = og;
            cell[row * H + j] = c;
            out[((int64_t)t * B + b) * D * H + d * H + j] = h;
            h_s[j] = h;
        }
        __syncthreads();
    }
}

// dout (T, B, D*H) -> dgx (T, B, D, 4H): gradient w.r.t. the gate pre-activations (what the projection GEMM consumes)
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_backward_kernel(const float *__restrict__ dout,
                                                              const float *__restrict__ w_hh,
                                                              const float *__restrict__ gates,
                                                              const float *__restrict__ cell, float *__restrict__ dgx,
                                                              int T, int B, int D) {
    __shared__ __attribute__((aligned(16))) float dg_s[4 * H];
    __shared__ float part[4 * H];
    __shared__ float dh_s[H];
    const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const int k = tid % H, q = tid / H;
    // thread (k, q) holds W_hh[q*H + jj][k], jj < H: its share of the transposed product dh[k] = sum_j dg[j] W_hh[j][k]
    float w[H];
    {
        const float *wc = w_hh + ((int64_t)d * 4 * H + (int64_t)q * H) * H + k;
#pragma unroll
        for (int jj = 0; jj < H; ++jj) w[jj] = wc[(int64_t)jj * H];
    }
    if (tid < H) dh_s[tid] = 0.0f;
    float dc_next = 0.0f;
    __syncthreads();
    for (int step = T - 1; step >= 0; --step) {
        const int t = d == 0 ? step : T - 1 - step;
        const int64_t row = ((int64_t)t * B + b) * D + d;
        if (tid < H) {
            const int j = tid;
            const float *gr = gates + row * 4 * H;
            const float ig = gr[j], fg = gr[H + j], gg = gr[2 * H + j], og = gr[3 * H + j];
            const float c = cell[row * H + j];
            float c_prev = 0.0f;
            if (step > 0) {
                const int tp = d == 0 ? step - 1 : T - step;
                c_prev = cell[(((int64_t)tp * B + b) * D + d) * H + j];
            }
            const float dh = dout[((int64_t)t * B + b) * D * H + d * H + j] + dh_s[j];
            const float tc = tanhf(c);
            const float d_o = dh * tc * og * (1.0f - og);
            const float dc = dh * og * (1.0f - tc * tc) + dc_next;
            const float d_i = dc * gg * ig * (1.0f - ig);
            const float d_f = dc * c_prev * fg * (1.0f - fg);
            const float d_g = dc * ig * (1.0f - gg * gg);
            dc_next = dc * fg;
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
        __syncthreads();
        float s = 0.0f;
#pragma unroll
        for (int jj = 0; jj < H; jj += 4) {
            const float4 gv = *reinterpret_cast<const float4 *>(&dg_s[q * H + jj]);
            s = fmaf(gv.x, w[jj], s);
            s = fmaf(gv.y, w[jj + 1], s);
            s = fmaf(gv.z, w[jj + 2], s);
            s = fmaf(gv.w, w[jj + 3], s);
        }
        part[tid] = s;
        __syncthreads();
        if (tid < H) dh_s[tid] = (part[tid] + part[H + tid]) + (part[2 * H + tid] + part[3 * H + tid]);
        __syncthreads();
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
                       w_hh, gates, cell, dgx, (int)T, (int)B, (int)D);
    return status_after_launch();
}

}  // extern "C"
