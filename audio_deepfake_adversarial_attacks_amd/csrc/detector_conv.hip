// detector_conv.hip — the two 3x3 convolutions at the spectrogram end of SpecRNet's first residual block, where one side
// of the convolution has only 1-2 channels (C ABI: include/advstep_detector.h):
//     forward   conv1: (N, Cin <= 2, H, W) -> (N, Cout, H, W), + shift, LeakyReLU        (src/models/specrnet.py:73-81, block0)
//     backward  d x = conv1^T(d conv1 out) + conv_downsample^T(d(conv2 out + identity))  : (N, K, H, W) -> (N, rows <= 2, H, W)
// A 16-row matrix tile would be 7/8 padding here (the Winograd kernel of lcnn_wino.hip needs 230 / 460 us for them at
// B = 128, 80 x 404); both are HBM streaming over the 20-channel activation (331 MB) with a few hundred FMAs per position,
// so they run on the vector ALUs: thread = one 2x2 block of positions, the small side (input window / accumulators) in
// registers, the loop over the large side with the 3x3 taps as wave-uniform scalar operands and v_pk_fma_f32 over
// horizontally adjacent positions.  Zero padding comes from the buffer descriptor (out-of-range offset reads 0); the
// channel offset of the loop travels in the scalar offset, so the loop has no address arithmetic.
// Accumulation order is fixed (channels, then taps row-major), fma contraction allowed: deterministic, within float
// rounding of ATen / MIOpen.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "advstep_detector.h"

namespace {

constexpr int kBlock = 256;
typedef float f32x2 __attribute__((ext_vector_type(2)));

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t as_stream(advstep_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int status_after_launch() { return hipGetLastError() == hipSuccess ? ADVSTEP_OK : ADVSTEP_ELAUNCH; }

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// byte offsets of the 4x4 window around the 2x2 block (th, tw) inside one (H, W) plane; 0x80000000 = outside (reads 0)
__device__ __forceinline__ void window_offsets(int th, int tw, int H, int W, bool valid, uint32_t (&off)[4][4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = 2 * th - 1 + p, w = 2 * tw - 1 + q;
            off[p][q] = (valid && h >= 0 && h < H && w >= 0 && w < W) ? (uint32_t)(h * W + w) * 4u : 0x80000000u;
        }
}

// ---- few input channels: y[co] = lrelu(sum_ci conv3x3(x[ci], w[co][ci]) + shift[co]) -------------------------------------
template <int CIN>
__global__ __launch_bounds__(kBlock) void fewin_forward_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ shift, float slope,
                                                               float *__restrict__ y, uint8_t *__restrict__ act, int Cout, int H,
                                                               int W) {
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    const int64_t n = blockIdx.y;
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = t < TH * TW;
    const int th = valid ? t / TW : 0, tw = valid ? t - th * TW : 0;
    const uint32_t plane = (uint32_t)(H * W);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + n * (int64_t)CIN * plane), 0,
                                                                        (int)(CIN * plane * 4u), 0x00020000);
    uint32_t off[4][4];
    window_offsets(th, tw, H, W, valid, off);
    // pairs of horizontally adjacent window entries at even (pe) and odd (po) column offsets: every packed operand is an
    // aligned register pair
    f32x2 pe[CIN][4][2], po[CIN][4];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = buf_load(xr, off[p][q], (uint32_t)ci * plane * 4u);
            pe[ci][p][0] = (f32x2){v[0], v[1]};
            pe[ci][p][1] = (f32x2){v[2], v[3]};
            po[ci][p] = (f32x2){v[1], v[2]};
        }
    if (!valid) return;
    const bool h1 = 2 * th + 1 < H, w1 = 2 * tw + 1 < W, even = (W & 1) == 0;
    float *yo = y + (n * (int64_t)Cout * H + 2 * th) * W + 2 * tw;
    for (int co = 0; co < Cout; ++co) {
        const float *wc = w + (int64_t)co * CIN * 9;          // wave-uniform address: scalar loads
        f32x2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};           // rows 0 / 1 of the 2x2 block
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float u = wc[ci * 9 + a * 3 + b];
                    const f32x2 t0 = b == 1 ? po[ci][a] : pe[ci][a][b >> 1];
                    const f32x2 t1 = b == 1 ? po[ci][a + 1] : pe[ci][a + 1][b >> 1];
                    r0 = __builtin_elementwise_fma((f32x2){u, u}, t0, r0);
                    r1 = __builtin_elementwise_fma((f32x2){u, u}, t1, r1);
                }
        const float sh = shift ? shift[co] : 0.0f;
        r0 += (f32x2){sh, sh};
        r1 += (f32x2){sh, sh};
        // the activation's sign bytes (N, Cout, TH, TW), bit 2 i + j = output (i, j) of this 2x2 block > 0: what the input
        // gradient of the NEXT convolution multiplies by (lcnn_wino.hip EPI 6) instead of reading y again; 0 outside an odd-sized plane
        if (act)
            act[((n * Cout + co) * TH + th) * (int64_t)TW + tw] =
                (uint8_t)((r0.x > 0.0f) | ((w1 && r0.y > 0.0f) << 1) | ((h1 && r1.x > 0.0f) << 2) | ((h1 && w1 && r1.y > 0.0f) << 3));
        r0.x = r0.x > 0.0f ? r0.x : r0.x * slope;
        r0.y = r0.y > 0.0f ? r0.y : r0.y * slope;
        r1.x = r1.x > 0.0f ? r1.x : r1.x * slope;
        r1.y = r1.y > 0.0f ? r1.y : r1.y * slope;
        float *o = yo + (int64_t)co * plane;
        if (even) {
            *reinterpret_cast<f32x2 *>(o) = r0;
            if (h1) *reinterpret_cast<f32x2 *>(o + W) = r1;
        } else {
            o[0] = r0.x;
            if (w1) o[1] = r0.y;
            if (h1) {
                o[W] = r1.x;
                if (w1) o[W + 1] = r1.y;
            }
        }
    }
}

// ---- few output rows: gx[r] = sum_k conv3x3(g1[k], rot180(w3[k][r])) + sum_k wd[k][r] * unpool(gp, sel)[k] ---------------
// w3 (K, ROWS, 3, 3) and wd (K, ROWS) are FORWARD weights of convolutions with ROWS input channels (this is their input
// gradient); gp (N, K, H/2, W/2) + sel: the pooled gradient of the identity path in compact form (may be NULL).
template <int ROWS>
__global__ __launch_bounds__(kBlock) void fewout_grad_kernel(const float *__restrict__ g1, const float *__restrict__ w3,
                                                             const float *__restrict__ gp, const uint8_t *__restrict__ sel,
                                                             const float *__restrict__ wd, float *__restrict__ gx, int K, int H,
                                                             int W) {
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1, Hs = H >> 1, Ws = W >> 1;
    const int64_t n = blockIdx.y;
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = t < TH * TW;
    const int th = valid ? t / TW : 0, tw = valid ? t - th * TW : 0;
    const uint32_t plane = (uint32_t)(H * W), cplane = (uint32_t)(Hs * Ws);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g1 + n * (int64_t)K * plane), 0,
                                                                        (int)((uint32_t)K * plane * 4u), 0x00020000);
    const bool pooled = gp != nullptr;
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pooled ? gp + n * (int64_t)K * cplane : g1), 0, pooled ? (int)((uint32_t)K * cplane * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(pooled ? sel + n * (int64_t)K * cplane : reinterpret_cast<const uint8_t *>(g1)), 0,
        pooled ? (int)((uint32_t)K * cplane) : 0, 0x00020000);
    uint32_t off[4][4];
    window_offsets(th, tw, H, W, valid, off);
    const uint32_t cell = (valid && th < Hs && tw < Ws) ? (uint32_t)(th * Ws + tw) : 0x20000000u;   // the block's pooling window

    f32x2 acc[ROWS][2];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r][0] = acc[r][1] = (f32x2){0.0f, 0.0f};
    // the window (and pooled cell) of channel k + 1 is requested before channel k's FMAs issue: the loop is otherwise one
    // memory latency per channel
    struct Taps {
        float v[4][4];
        float gv;
        uint32_t code;
    };
    auto request = [&](Taps &t, int k) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) t.v[p][q] = buf_load(gr, off[p][q], (uint32_t)k * plane * 4u);
        if (pooled) {
            t.gv = buf_load(pr, cell << 2, (uint32_t)k * cplane * 4u);
            t.code = __builtin_amdgcn_raw_buffer_load_b8(sr, cell, (uint32_t)k * cplane, 0);
        }
    };
    auto consume = [&](const Taps &t, int k) {
        f32x2 pe[4][2], po[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            pe[p][0] = (f32x2){t.v[p][0], t.v[p][1]};
            pe[p][1] = (f32x2){t.v[p][2], t.v[p][3]};
            po[p] = (f32x2){t.v[p][1], t.v[p][2]};
        }
        const float *wk = w3 + (int64_t)k * ROWS * 9;         // wave-uniform: scalar loads
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float u = wk[r * 9 + (2 - a) * 3 + (2 - b)];
                    const f32x2 t0 = b == 1 ? po[a] : pe[a][b >> 1];
                    const f32x2 t1 = b == 1 ? po[a + 1] : pe[a + 1][b >> 1];
                    acc[r][0] = __builtin_elementwise_fma((f32x2){u, u}, t0, acc[r][0]);
                    acc[r][1] = __builtin_elementwise_fma((f32x2){u, u}, t1, acc[r][1]);
                }
        if (pooled) {
            const f32x2 d0 = {t.code == 0u ? t.gv : 0.0f, t.code == 1u ? t.gv : 0.0f};
            const f32x2 d1 = {t.code == 2u ? t.gv : 0.0f, t.code == 3u ? t.gv : 0.0f};
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const float u = wd[(int64_t)k * ROWS + r];
                acc[r][0] = __builtin_elementwise_fma((f32x2){u, u}, d0, acc[r][0]);
                acc[r][1] = __builtin_elementwise_fma((f32x2){u, u}, d1, acc[r][1]);
            }
        }
    };
    Taps ta, tb;
    request(ta, 0);
    int k = 0;
    for (; k + 1 < K; k += 2) {
        request(tb, k + 1);
        consume(ta, k);
        if (k + 2 < K) request(ta, k + 2);
        consume(tb, k + 1);
    }
    if (k < K) consume(ta, k);
    if (!valid) return;
    const bool h1 = 2 * th + 1 < H, w1 = 2 * tw + 1 < W, even = (W & 1) == 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float *o = gx + ((n * ROWS + r) * (int64_t)H + 2 * th) * W + 2 * tw;
        if (even) {
            *reinterpret_cast<f32x2 *>(o) = acc[r][0];
            if (h1) *reinterpret_cast<f32x2 *>(o + W) = acc[r][1];
        } else {
            o[0] = acc[r][0].x;
            if (w1) o[1] = acc[r][0].y;
            if (h1) {
                o[W] = acc[r][1].x;
                if (w1) o[W + 1] = acc[r][1].y;
            }
        }
    }
}

// The same gradient with a thread owning a 2x4 block of positions (W % 4 == 0): a window row is one 16-byte load (columns
// 4 tq .. 4 tq + 3, contiguous across lanes) plus the two edge columns, 12 load instructions per channel for 8 outputs instead of
// 32 — the 2x2 version is bound by the texture path (lanes 8 bytes apart, 16 dword loads per channel and tile), not by HBM.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROWS>
__global__ __launch_bounds__(kBlock) void fewout_grad4_kernel(const float *__restrict__ g1, const float *__restrict__ w3,
                                                              const float *__restrict__ gp, const uint8_t *__restrict__ sel,
                                                              const float *__restrict__ wd, float *__restrict__ gx, int K, int H,
                                                              int W) {
    const int TH = (H + 1) >> 1, TQ = W >> 2, Hs = H >> 1, Ws = W >> 1;
    const int64_t n = blockIdx.y;
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = t < TH * TQ;
    const int th = valid ? t / TQ : 0, tq = valid ? t - th * TQ : 0;
    const uint32_t plane = (uint32_t)(H * W), cplane = (uint32_t)(Hs * Ws);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g1 + n * (int64_t)K * plane), 0,
                                                                        (int)((uint32_t)K * plane * 4u), 0x00020000);
    const bool pooled = gp != nullptr;
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pooled ? gp + n * (int64_t)K * cplane : g1), 0, pooled ? (int)((uint32_t)K * cplane * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(pooled ? sel + n * (int64_t)K * cplane : reinterpret_cast<const uint8_t *>(g1)), 0,
        pooled ? (int)((uint32_t)K * cplane) : 0, 0x00020000);
    // per window row: byte offset of the aligned quad, of the column left of it and of the column right of it
    uint32_t quad[4], left[4], right[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int h = 2 * th - 1 + p;
        const bool row = valid && h >= 0 && h < H;
        quad[p] = row ? (uint32_t)(h * W + 4 * tq) * 4u : 0x80000000u;
        left[p] = (row && tq > 0) ? quad[p] - 4u : 0x80000000u;
        right[p] = (row && 4 * tq + 4 < W) ? quad[p] + 16u : 0x80000000u;
    }
    const bool in_pool = valid && th < Hs;
    const uint32_t cell = in_pool ? (uint32_t)(th * Ws + 2 * tq) : 0x20000000u;       // the block's two pooling windows

    f32x2 acc[ROWS][2][2];                                   // [row of gx][block row][column pair]
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[r][i][0] = acc[r][i][1] = (f32x2){0.0f, 0.0f};
    for (int k = 0; k < K; ++k) {
        const uint32_t soff = (uint32_t)k * plane * 4u;
        f32x2 ev[4][3], od[4][2];                            // (v0,v1) (v2,v3) (v4,v5)  /  (v1,v2) (v3,v4), v0 = column 4 tq - 1
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x4 q4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, quad[p], soff, 0));
            const float l = buf_load(gr, left[p], soff), rr = buf_load(gr, right[p], soff);
            ev[p][0] = (f32x2){l, q4.x};
            ev[p][1] = (f32x2){q4.y, q4.z};
            ev[p][2] = (f32x2){q4.w, rr};
            od[p][0] = (f32x2){q4.x, q4.y};
            od[p][1] = (f32x2){q4.z, q4.w};
        }
        const float *wk = w3 + (int64_t)k * ROWS * 9;         // wave-uniform: scalar loads
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float u = wk[r * 9 + (2 - a) * 3 + (2 - b)];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            // outputs (2c, 2c + 1) of block row i read window columns 2c + b, 2c + b + 1 of window row i + a
                            const f32x2 tp = b == 1 ? od[i + a][c] : ev[i + a][c + (b >> 1)];
                            acc[r][i][c] = __builtin_elementwise_fma((f32x2){u, u}, tp, acc[r][i][c]);
                        }
                }
        if (pooled) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float gv = buf_load(pr, (cell + (uint32_t)c) << 2, (uint32_t)k * cplane * 4u);
                const uint32_t code = __builtin_amdgcn_raw_buffer_load_b8(sr, cell + (uint32_t)c, (uint32_t)k * cplane, 0);
                const f32x2 d0 = {code == 0u ? gv : 0.0f, code == 1u ? gv : 0.0f};
                const f32x2 d1 = {code == 2u ? gv : 0.0f, code == 3u ? gv : 0.0f};
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const float u = wd[(int64_t)k * ROWS + r];
                    acc[r][0][c] = __builtin_elementwise_fma((f32x2){u, u}, d0, acc[r][0][c]);
                    acc[r][1][c] = __builtin_elementwise_fma((f32x2){u, u}, d1, acc[r][1][c]);
                }
            }
        }
    }
    if (!valid) return;
    const bool h1 = 2 * th + 1 < H;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float *o = gx + ((n * ROWS + r) * (int64_t)H + 2 * th) * W + 4 * tq;
        *reinterpret_cast<f32x4 *>(o) = (f32x4){acc[r][0][0].x, acc[r][0][0].y, acc[r][0][1].x, acc[r][0][1].y};
        if (h1) *reinterpret_cast<f32x4 *>(o + W) = (f32x4){acc[r][1][0].x, acc[r][1][0].y, acc[r][1][1].x, acc[r][1][1].y};
    }
}

bool dims_ok(int64_t N, int64_t big, int64_t H, int64_t W) {
    // one sample's planes are addressed through a 32-bit buffer descriptor; samples ride on blockIdx.y
    return N >= 0 && N <= 65535 && H >= 0 && W >= 0 && big >= 1 && (uint64_t)big * H * W * 4 < (1ull << 31);
}

}  // namespace

extern "C" {

int advstep_conv3x3_fewin_supported(int64_t Cin) { return Cin == 1 || Cin == 2; }

int advstep_conv3x3_fewin_forward_act_f32(const float *x, const float *w, const float *shift, float slope, float *y,
                                          uint8_t *act, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                                          advstep_stream_t stream) {
    if (!advstep_conv3x3_fewin_supported(Cin) || !dims_ok(N, Cout, H, W)) return ADVSTEP_EINVAL;
    if (N * H * W == 0) return ADVSTEP_OK;
    if (!x || !w || !y) return ADVSTEP_EINVAL;
    const dim3 grid((unsigned)ceil_div(((H + 1) / 2) * ((W + 1) / 2), kBlock), (unsigned)N), block(kBlock);
    if (Cin == 1)
        hipLaunchKernelGGL(fewin_forward_kernel<1>, grid, block, 0, as_stream(stream), x, w, shift, slope, y, act, (int)Cout, (int)H, (int)W);
    else
        hipLaunchKernelGGL(fewin_forward_kernel<2>, grid, block, 0, as_stream(stream), x, w, shift, slope, y, act, (int)Cout, (int)H, (int)W);
    return status_after_launch();
}

int advstep_conv3x3_fewin_forward_f32(const float *x, const float *w, const float *shift, float slope, float *y, int64_t N,
                                      int64_t Cin, int64_t Cout, int64_t H, int64_t W, advstep_stream_t stream) {
    return advstep_conv3x3_fewin_forward_act_f32(x, w, shift, slope, y, nullptr, N, Cin, Cout, H, W, stream);
}

int advstep_conv3x3_fewout_grad_f32(const float *g1, const float *w3, const float *gp, const uint8_t *sel, const float *wd,
                                    float *gx, int64_t N, int64_t K, int64_t rows, int64_t H, int64_t W,
                                    advstep_stream_t stream) {
    if (!advstep_conv3x3_fewin_supported(rows) || !dims_ok(N, K, H, W)) return ADVSTEP_EINVAL;
    if (N * H * W == 0) return ADVSTEP_OK;
    if (!g1 || !w3 || !gx || ((gp != nullptr) != (sel != nullptr)) || (gp && !wd)) return ADVSTEP_EINVAL;
    if ((H / 2) * (W / 2) == 0) gp = nullptr, sel = nullptr;       // nothing was pooled: the identity path has no gradient
    // ADVSTEP_FEWOUT_QUADS=0 (read per call) keeps the 2x2-block kernel for every shape (A/B measurements)
    const char *e = getenv("ADVSTEP_FEWOUT_QUADS");
    if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(g1) & 15u) == 0 && (reinterpret_cast<uintptr_t>(gx) & 15u) == 0 &&
        !(e && e[0] == '0')) {
        const dim3 grid4((unsigned)ceil_div(((H + 1) / 2) * (W / 4), kBlock), (unsigned)N), block4(kBlock);
        if (rows == 1)
            hipLaunchKernelGGL(fewout_grad4_kernel<1>, grid4, block4, 0, as_stream(stream), g1, w3, gp, sel, wd, gx, (int)K, (int)H, (int)W);
        else
            hipLaunchKernelGGL(fewout_grad4_kernel<2>, grid4, block4, 0, as_stream(stream), g1, w3, gp, sel, wd, gx, (int)K, (int)H, (int)W);
        return status_after_launch();
    }
    const dim3 grid((unsigned)ceil_div(((H + 1) / 2) * ((W + 1) / 2), kBlock), (unsigned)N), block(kBlock);
    if (rows == 1)
        hipLaunchKernelGGL(fewout_grad_kernel<1>, grid, block, 0, as_stream(stream), g1, w3, gp, sel, wd, gx, (int)K, (int)H, (int)W);
    else
        hipLaunchKernelGGL(fewout_grad_kernel<2>, grid, block, 0, as_stream(stream), g1, w3, gp, sel, wd, gx, (int)K, (int)H, (int)W);
    return status_after_launch();
}

}  // extern "C"
