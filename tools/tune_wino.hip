// Prototype / tuning harness: 3x3 convolution (pad 1) + bias + max-feature-map + 2x2 max-pool as ONE kernel,
// Winograd F(2x2, 3x3) with the 16 per-position GEMMs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/tune_wino.hip -o tools/_tune_wino.bin
//   tools/_tune_wino.bin [N CIN COUT H W]        (default: LCNN layer 6, 128 32 96 202 40)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool takes_b(float a, float b) { return !(a != a) && !(a >= b); }
__device__ __forceinline__ float pool_select(float a00, float b00, float a01, float b01, float a10, float b10, float a11,
                                             float b11, int &code) {
    const bool t00 = takes_b(a00, b00), t01 = takes_b(a01, b01), t10 = takes_b(a10, b10), t11 = takes_b(a11, b11);
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

constexpr int kWaves = 8, kThreads = kWaves * 64;

// U: [slice][xi][cin][16][2] (pair = the two channel halves of MFM channel slice*16 + j).
// 1-D grid of R * slices persistent workgroups, R a multiple of 8.  Workgroup i runs on XCD i % 8; the `slices`
// workgroups that walk the same tile range (and so read the same input) get ids that differ by 8: same XCD, same L2.
template <int CIN, int ABL>
__global__ __launch_bounds__(kThreads) void wino_fwd(const float *__restrict__ x, const float *__restrict__ U,
                                                     const float *__restrict__ bias, float *__restrict__ y,
                                                     uint8_t *__restrict__ idx, int N, int H, int W, int C2, int slices) {
    extern __shared__ __attribute__((aligned(16))) float u_s[];
    constexpr int kSlice = 16 * CIN * 32, S = CIN / 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slice = j % slices, range = (j / slices) * 8 + xcd, ranges = gridDim.x / slices;
    {
        const float4 *src = reinterpret_cast<const float4 *>(U + (size_t)slice * kSlice);
        float4 *dst = reinterpret_cast<float4 *>(u_s);
        for (int i = threadIdx.x; i < kSlice / 4; i += kThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, nl = lane & 15;
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1, Ho = H >> 1, Wo = W >> 1;
    const int tiles = N * TH * TW, groups = (tiles + 15) >> 4;
    const uint32_t plane = (uint32_t)(H * W);
    // raw buffer over x: an out-of-range offset reads as 0 — that is the convolution's zero padding, for free
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0,
                                                                       (int)((size_t)N * CIN * plane * 4), 0x00020000);
    for (int grp = range * kWaves + wave; grp < groups; grp += ranges * kWaves) {
        const int t = grp * 16 + nl;
        const bool valid = t < tiles;
        const int tt = valid ? t : tiles - 1;
        const int n = tt / (TH * TW), rem = tt - n * (TH * TW), th = rem / TW, tw = rem - th * TW;
        uint32_t voff[4][4];
        const uint32_t lane_base = ((uint32_t)(n * CIN + g)) * plane;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hh = 2 * th - 1 + p, ww = 2 * tw - 1 + q;
                const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                voff[p][q] = ok ? (lane_base + (uint32_t)(hh * W + ww)) * 4u : 0x80000000u;
            }
        f32x4 acc[16][2];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[xi][0] = acc[xi][1] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

        auto load_patch = [&](float (&dst)[4][4], int s) {
            const uint32_t soff = (uint32_t)(4 * s) * plane * 4u;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[p][q] = ABL == 1 ? (float)(p + q + s) : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff[p][q], soff, 0));
        };
        auto step = [&](const float (&d)[4][4], int s) {
            // V = B^T d B
            float tr[4][4], v[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tr[0][q] = d[0][q] - d[2][q];
                tr[1][q] = d[1][q] + d[2][q];
                tr[2][q] = d[2][q] - d[1][q];
                tr[3][q] = d[1][q] - d[3][q];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                v[a][0] = tr[a][0] - tr[a][2];
                v[a][1] = tr[a][1] + tr[a][2];
                v[a][2] = tr[a][2] - tr[a][1];
                v[a][3] = tr[a][1] - tr[a][3];
            }
            const float *us = u_s + ((4 * s + g) * 16 + nl) * 2;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                const f32x2 a = *reinterpret_cast<const f32x2 *>(us + xi * (CIN * 32));
                if (ABL == 2) {
                    acc[xi][0][0] += a.x * v[xi >> 2][xi & 3];
                    acc[xi][1][0] += a.y * v[xi >> 2][xi & 3];
                } else {
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, v[xi >> 2][xi & 3], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, v[xi >> 2][xi & 3], acc[xi][1], 0, 0, 0);
                }
            }
        };
        float da[4][4], db[4][4];
        load_patch(da, 0);
#pragma unroll 1
        for (int s = 0; s < S; s += 2) {
            load_patch(db, s + 1);            // in flight while this step's MFMAs run
            step(da, s);
            if (s + 2 < S) load_patch(da, s + 2);
            step(db, s + 1);
        }
        // epilogue: Y = A^T M A per (channel, tile), bias, max-feature-map, 2x2 pool
        const bool store = valid && th < Ho && tw < Wo;
        if (ABL == 3) {
            float sum = 0.0f;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) sum += acc[xi][0][0] + acc[xi][1][1] + acc[xi][0][2] + acc[xi][1][3];
            if (store) y[((size_t)n * C2 + slice * 16 + 4 * g) * Ho * Wo + (size_t)th * Wo + tw] = sum;
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = slice * 16 + 4 * g + r;
            float yy[2][2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float s0[4], s1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    s0[b] = acc[0 + b][m][r] + acc[4 + b][m][r] + acc[8 + b][m][r];
                    s1[b] = acc[4 + b][m][r] - acc[8 + b][m][r] - acc[12 + b][m][r];
                }
                const float bs = bias[m * C2 + ch];
                yy[m][0][0] = s0[0] + s0[1] + s0[2] + bs;
                yy[m][0][1] = s0[1] - s0[2] - s0[3] + bs;
                yy[m][1][0] = s1[0] + s1[1] + s1[2] + bs;
                yy[m][1][1] = s1[1] - s1[2] - s1[3] + bs;
            }
            int code;
            const float vbest = pool_select(yy[0][0][0], yy[1][0][0], yy[0][0][1], yy[1][0][1], yy[0][1][0], yy[1][1][0],
                                            yy[0][1][1], yy[1][1][1], code);
            if (store) {
                const size_t o = ((size_t)n * C2 + ch) * Ho * Wo + (size_t)th * Wo + tw;
                y[o] = vbest;
                idx[o] = (uint8_t)code;
            }
        }
    }
}

#define LAUNCH1(C, A) do { CK(hipFuncSetAttribute((const void *)wino_fwd<C, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((wino_fwd<C, A>), grid, dim3(kThreads), lds, 0, dx, dU, db, dy, di, N, H, W, C2, slices); } while (0)
#define LAUNCH(C) do { if (abl == 0) LAUNCH1(C, 0); else if (abl == 1) LAUNCH1(C, 1); else if (abl == 2) LAUNCH1(C, 2); else LAUNCH1(C, 3); } while (0)

static void host_U(const std::vector<float> &w, int CIN, int COUT, std::vector<float> &U) {
    const int C2 = COUT / 2, slices = C2 / 16;
    const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    U.assign((size_t)slices * 16 * CIN * 32, 0.0f);
    for (int co = 0; co < COUT; ++co)
        for (int ci = 0; ci < CIN; ++ci) {
            double gg[3][3], t[4][3], u[4][4];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gg[i][j] = w[((size_t)co * CIN + ci) * 9 + i * 3 + j];
            for (int a = 0; a < 4; ++a) for (int j = 0; j < 3; ++j) { t[a][j] = 0; for (int i = 0; i < 3; ++i) t[a][j] += G[a][i] * gg[i][j]; }
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { u[a][b] = 0; for (int j = 0; j < 3; ++j) u[a][b] += t[a][j] * G[b][j]; }
            const int half = co >= C2, c = co - half * C2, slice = c / 16, j = c % 16;
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b)
                U[((((size_t)slice * 16 + (a * 4 + b)) * CIN + ci) * 16 + j) * 2 + half] = (float)u[a][b];
        }
}

int main(int argc, char **argv) {
    int N = 128, CIN = 32, COUT = 96, H = 202, W = 40;
    if (argc >= 6) { N = atoi(argv[1]); CIN = atoi(argv[2]); COUT = atoi(argv[3]); H = atoi(argv[4]); W = atoi(argv[5]); }
    const int C2 = COUT / 2, Ho = H / 2, Wo = W / 2, slices = C2 / 16;
    std::vector<float> x((size_t)N * CIN * H * W), w((size_t)COUT * CIN * 9), bias(COUT), U;
    srand(1);
    for (auto &v : x) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
    for (auto &v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (auto &v : bias) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    host_U(w, CIN, COUT, U);
    float *dx, *dU, *db, *dy;
    uint8_t *di;
    const size_t ny = (size_t)N * C2 * Ho * Wo;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dU, U.size() * 4)); CK(hipMalloc(&db, bias.size() * 4));
    CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&di, ny));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)16 * CIN * 32 * 4;
    int abl = argc >= 7 ? atoi(argv[6]) : 0;
    auto launch = [&](int blocks) {
        dim3 grid(blocks * slices);
        switch (CIN) {
            case 32: LAUNCH(32); break;
            case 48: LAUNCH(48); break;
            case 64: LAUNCH(64); break;
            default: printf("unsupported CIN\n"); exit(1);
        }
        CK(hipGetLastError());
    };
    const int blocks = (256 / slices) / 8 * 8;   // tile ranges: a multiple of 8 (XCD-aligned), ranges * slices <= 256 CUs
    launch(blocks);
    CK(hipDeviceSynchronize());
    // check n = 0 and n = N-1, a few channels, against direct convolution in double
    std::vector<float> y(ny);
    std::vector<uint8_t> id(ny);
    CK(hipMemcpy(y.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(id.data(), di, ny, hipMemcpyDeviceToHost));
    double maxerr = 0;
    long bad_code = 0, checked = 0;
    for (int n : {0, N - 1})
        for (int ch : {0, 5, 17, C2 - 1})
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    double best = -1e300;
                    int code = 0;
                    double vals[2][4];
                    for (int pos = 0; pos < 4; ++pos) {
                        const int h = 2 * ho + (pos >> 1), ww = 2 * wo + (pos & 1);
                        for (int m = 0; m < 2; ++m) {
                            const int co = m * C2 + ch;
                            double s = bias[co];
                            for (int ci = 0; ci < CIN; ++ci)
                                for (int kh = 0; kh < 3; ++kh)
                                    for (int kw = 0; kw < 3; ++kw) {
                                        const int hh = h + kh - 1, w2 = ww + kw - 1;
                                        if (hh < 0 || hh >= H || w2 < 0 || w2 >= W) continue;
                                        s += (double)w[((size_t)co * CIN + ci) * 9 + kh * 3 + kw] * x[(((size_t)n * CIN + ci) * H + hh) * W + w2];
                                    }
                            vals[m][pos] = s;
                        }
                        const bool tb = !(vals[0][pos] >= vals[1][pos]);
                        const double mv = tb ? vals[1][pos] : vals[0][pos];
                        if (mv > best) { best = mv; code = (tb << 2) | pos; }
                    }
                    const size_t o = ((size_t)n * C2 + ch) * Ho * Wo + (size_t)ho * Wo + wo;
                    maxerr = fmax(maxerr, fabs(best - y[o]));
                    bad_code += id[o] != code;
                    ++checked;
                }
    printf("check: %ld outputs, max abs err %.3e, selection codes differing %ld\n", checked, maxerr, bad_code);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blk : {blocks, 2 * blocks}) {
        launch(blk);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch(blk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 100.0;
        const double direct = 2.0 * CIN * 9 * COUT * (double)N * H * W;
        printf("grid (%d, %d): %.1f us  = %.1f TFLOP/s direct-conv equivalent, %.1f TFLOP/s on the matrix cores\n", blk,
               slices, us, direct / us * 1e-6, direct / 2.25 / us * 1e-6);
    }
    return 0;
}
