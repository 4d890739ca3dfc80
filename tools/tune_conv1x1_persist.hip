// Harness (round 6): conv1x1_mfm_forward / backward at LCNN's first 1x1 block (N = 128, Cin = 32, C = 32, P = 8080), the
// shipped one-tile-per-wave structure against
//   * the same kernel with its selection words gone (timing only), or written as ONE coalesced store per wave
//     (a lane keeps the bits of its own 16 channels: [tile][lane] uint16),
//   * a PERSISTENT grid (G workgroups per CU, W staged once) whose waves walk their tiles with the fragments of the
//     next D - 1 tiles in flight in registers (buffer loads, out-of-range tiles read 0: no load sits inside a branch),
//   * non-temporal loads / stores.
// Every variant's y is compared with the first one's bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/tune_conv1x1_persist.hip -o tools/_tune_c11q.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bool takes_b(float a, float b) { return !(a != a) && !(a >= b); }
constexpr uint32_t kOut = 0x80000000u;
constexpr int CIN = 32, CP = 32, PITCH = CIN + 1;

// a zero the compiler cannot see through: added to an LDS address inside the tile loop it keeps loop-invariant LDS reads
// (the 64 epilogue parameters, optionally the 32 W fragments) from being hoisted into registers
__device__ __forceinline__ int opaque_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }

__device__ __forceinline__ void stage(const float *__restrict__ weight, const float *__restrict__ par, float *w_s) {
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = weight[threadIdx.x + j * 256];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = threadIdx.x + j * 256, row = i / CIN, ci = i - row * CIN;
        w_s[row * PITCH + ci] = wv[j];
    }
    if (threadIdx.x < 128) w_s[2 * CP * PITCH + threadIdx.x] = par[threadIdx.x];
    __syncthreads();
}

// SEL 0: shipped words [n][c][p / 32]; 1: none (timing only); 2: lane masks [n][tile][pixel][half] uint16
// WREG: the 32 A fragments of the wave come from registers (wr) instead of LDS
template <int SEL, int AUX, bool WREG>
__device__ __forceinline__ void tile_compute(const float (&xb)[CIN / 2], const float *w_s, const float (&wr)[CIN], const float *par,
                                             __amdgpu_buffer_rsrc_t yr,
                                             __amdgpu_buffer_rsrc_t sr, uint32_t y_off, uint32_t y_soff, uint32_t s_off,
                                             uint32_t s_soff, uint32_t Pb, uint32_t PWb, bool valid, int li, int lk) {
    const float *wa = w_s + li * PITCH + lk;
    const float *wb = wa + CP * PITCH;
    f32x16 acc_a = {0}, acc_b = {0};
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) {
        acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(WREG ? wr[s] : wa[2 * s], xb[s], acc_a, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(WREG ? wr[16 + s] : wb[2 * s], xb[s], acc_b, 0, 0, 0);
    }
    uint32_t mask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c_lo = (r & 3) + 8 * (r >> 2), c = c_lo + 4 * lk;
        const float va = acc_a[r] + par[c], vb = acc_b[r] + par[CP + c];
        const bool tb = takes_b(va, vb);
        const float v = ((tb ? vb : va) - par[2 * CP + c]) * par[3 * CP + c];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, y_off, y_soff + (uint32_t)c_lo * Pb, AUX);
        if (SEL == 0) {
            const unsigned long long word = __ballot(valid && tb);
            const uint32_t sw = lk ? (uint32_t)(word >> 32) : (uint32_t)word;
            __builtin_amdgcn_raw_buffer_store_b32(sw, sr, s_off, s_soff + (uint32_t)c_lo * PWb, 0);
        } else if (SEL == 2) {
            mask |= tb ? (1u << r) : 0u;
        }
    }
    if (SEL == 2) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)mask, sr, s_off, s_soff, 0);
}

template <int SEL, int AUX>
__global__ __launch_bounds__(256) void k_base(const float *__restrict__ x, const float *__restrict__ weight,
                                              const float *__restrict__ par, float *__restrict__ y,
                                              uint32_t *__restrict__ sel, int N, int P, int PW) {
    extern __shared__ float w_s[];
    stage(weight, par, w_s);
    const int n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    if (tile * 32 >= P) return;
    const int p = tile * 32 + li;
    const bool valid = p < P;
    const uint32_t Pb = (uint32_t)P * 4u, PWb = (uint32_t)PW * 4u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)((size_t)N * CIN * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((size_t)N * CP * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(sel, 0, (int)((size_t)N * CP * PWb), 0x00020000);
    const uint32_t pb = (uint32_t)p * 4u;
    const uint32_t x_off = valid ? (uint32_t)lk * Pb + pb : kOut;
    const uint32_t y_off = valid ? 4u * (uint32_t)lk * Pb + pb : kOut;
    uint32_t s_off, s_soff;
    if (SEL == 2) {
        s_off = valid ? (uint32_t)(li * 2 + lk) * 2u : kOut;
        s_soff = (uint32_t)(n * PW + tile) * 128u;
    } else {
        s_off = li == 0 ? 4u * (uint32_t)lk * PWb + (uint32_t)tile * 4u : kOut;
        s_soff = (uint32_t)n * CP * PWb;
    }
    const uint32_t xs = (uint32_t)n * CIN * Pb, ys = (uint32_t)n * CP * Pb;
    float xb[CIN / 2];
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s)
        xb[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, x_off, xs + (uint32_t)(2 * s) * Pb, AUX));
    const float none[CIN] = {0};
    tile_compute<SEL, AUX, false>(xb, w_s, none, w_s + 2 * CP * PITCH, yr, sr, y_off, ys, s_off, s_soff, Pb, PWb, valid, li, lk);
}

// persistent: grid = CUs * G workgroups of 4 waves; wave w of the launch takes tiles w, w + W, w + 2 W, ...
template <int D, int SEL, int AUX, int G, bool HOISTW>
__global__ __launch_bounds__(256, G) void k_persist(const float *__restrict__ x, const float *__restrict__ weight,
                                                 const float *__restrict__ par, float *__restrict__ y,
                                                 uint32_t *__restrict__ sel, int N, int P, int PW, int iters) {
    extern __shared__ float w_s[];
    stage(weight, par, w_s);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const int nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
    float wr[CIN];
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) {
        wr[s] = HOISTW ? w_s[li * PITCH + lk + 2 * s] : 0.0f;
        wr[16 + s] = HOISTW ? w_s[(CP + li) * PITCH + lk + 2 * s] : 0.0f;
        if (HOISTW) { asm volatile("" : "+v"(wr[s])); asm volatile("" : "+v"(wr[16 + s])); }
    }
    const uint32_t Pb = (uint32_t)P * 4u, PWb = (uint32_t)PW * 4u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)((size_t)N * CIN * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((size_t)N * CP * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(sel, 0, (int)((size_t)N * CP * PWb), 0x00020000);
    const int dn = nwaves / PW, dt = nwaves - dn * PW;   // a step of nwaves tiles in (sample, tile) coordinates
    // load cursor and compute cursor: (sample, tile of the sample); the compute cursor lags D - 1 tiles
    int ln = __builtin_amdgcn_readfirstlane(w0 / PW), lt = __builtin_amdgcn_readfirstlane(w0 % PW);
    int cn = ln, ct = lt;
    auto advance = [&](int &n, int &t) {
        n += dn; t += dt;
        if (t >= PW) { t -= PW; ++n; }
    };
    auto issue = [&](float (&xb)[CIN / 2]) {
        const int p = lt * 32 + li;
        const bool valid = p < P && ln < N;
        const uint32_t x_off = valid ? (uint32_t)lk * Pb + (uint32_t)p * 4u : kOut;
        const uint32_t xs = (uint32_t)ln * CIN * Pb;
#pragma unroll
        for (int s = 0; s < CIN / 2; ++s)
            xb[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, x_off, xs + (uint32_t)(2 * s) * Pb, AUX));
        advance(ln, lt);
    };
    auto compute = [&](const float (&xb)[CIN / 2]) {
        const int p = ct * 32 + li;
        const bool valid = p < P && cn < N;
        const uint32_t y_off = valid ? 4u * (uint32_t)lk * Pb + (uint32_t)p * 4u : kOut;
        uint32_t s_off, s_soff;
        if (SEL == 2) {
            s_off = valid ? (uint32_t)(li * 2 + lk) * 2u : kOut;
            s_soff = (uint32_t)(cn * PW + ct) * 128u;
        } else {
            s_off = (li == 0 && cn < N) ? 4u * (uint32_t)lk * PWb + (uint32_t)ct * 4u : kOut;
            s_soff = (uint32_t)cn * CP * PWb;
        }
        tile_compute<SEL, AUX, HOISTW>(xb, w_s + opaque_zero(), wr, w_s + 2 * CP * PITCH + opaque_zero(), yr, sr, y_off,
                               (uint32_t)cn * CP * Pb, s_off, s_soff, Pb, PWb, valid, li, lk);
        advance(cn, ct);
    };
    float xb[D][CIN / 2];
#pragma unroll
    for (int j = 0; j < D - 1; ++j) issue(xb[j]);
#pragma unroll 1
    for (int k = 0; k < iters; k += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            issue(xb[(j + D - 1) % D]);
            compute(xb[j]);
        }
    }
}

// ---- backward: gx (Cin x P) = W^T (Cin x 2C) . routed gy ---------------------------------------------------------------
// SEL 0: shipped words, 16 broadcast loads per lane; 2: lane masks, one 4-byte load per lane (both halves' uint16)
template <int SEL, int AUX>
__device__ __forceinline__ void bwd_issue(__amdgpu_buffer_rsrc_t gr, __amdgpu_buffer_rsrc_t sr, uint32_t g_off, uint32_t g_soff,
                                          uint32_t s_off, uint32_t s_soff, uint32_t Pb, uint32_t PWb, float (&g)[16],
                                          uint32_t (&sw)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
        g[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, g_off, g_soff + (uint32_t)(2 * s) * Pb, AUX));
    if (SEL == 0) {
#pragma unroll
        for (int s = 0; s < 16; ++s) sw[s] = __builtin_amdgcn_raw_buffer_load_b32(sr, s_off, s_soff + (uint32_t)(2 * s) * PWb, 0);
    } else {
        sw[0] = __builtin_amdgcn_raw_buffer_load_b32(sr, s_off, s_soff, 0);
    }
}

template <int SEL, int AUX, bool WREG>
__device__ __forceinline__ void bwd_compute(const float (&g)[16], const uint32_t (&sw)[16], const float *w_s, const float (&wr)[32], const float *gs_s,
                                            __amdgpu_buffer_rsrc_t xr, uint32_t x_off, uint32_t x_soff, uint32_t Pb, int li,
                                            int lk) {
    constexpr int BP = 33;
    f32x16 acc = {0};
    const uint32_t m = SEL == 2 ? sw[0] >> lk : 0u;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int c = 2 * s + lk;
        const float gs = g[s] * gs_s[c];
        bool tb;
        if (SEL == 0) tb = (sw[s] >> li) & 1u;
        else {
            // channel c = 2 s + lk of the tile: forward lane half h = (s >> 1) & 1, accumulator register r = 2 (s & 1) + lk + 4 (s >> 2)
            const int bit = 16 * ((s >> 1) & 1) + 2 * (s & 1) + 4 * (s >> 2);
            tb = (m >> bit) & 1u;
        }
        const float ga = tb ? 0.0f : gs, gb = tb ? gs : 0.0f;
        const float *wa = w_s + c * BP + li;
        const float *wb = wa + CP * BP;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WREG ? wr[s] : wa[0], ga, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WREG ? wr[16 + s] : wb[0], gb, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ci_lo = (r & 3) + 8 * (r >> 2);
        const float gv = acc[r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gv), xr, x_off, x_soff + (uint32_t)ci_lo * Pb, AUX);
    }
}

__device__ __forceinline__ void stage_bwd(const float *__restrict__ weight, const float *__restrict__ gscale, float *w_s) {
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = weight[threadIdx.x + j * 256];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = threadIdx.x + j * 256, row = i / 32, ci = i - row * 32;
        w_s[row * 33 + ci] = wv[j];
    }
    if (threadIdx.x < 32) w_s[2 * CP * 33 + threadIdx.x] = gscale[threadIdx.x];
    __syncthreads();
}

template <int SEL, int AUX>
__global__ __launch_bounds__(256) void k_bwd_base(const float *__restrict__ gy, const uint32_t *__restrict__ sel,
                                                  const float *__restrict__ weight, const float *__restrict__ gscale,
                                                  float *__restrict__ gx, int N, int P, int PW) {
    extern __shared__ float w_s[];
    stage_bwd(weight, gscale, w_s);
    const int n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    if (tile * 32 >= P) return;
    const int p = tile * 32 + li;
    const bool valid = p < P;
    const uint32_t Pb = (uint32_t)P * 4u, PWb = (uint32_t)PW * 4u;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gy), 0, (int)((size_t)N * CP * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(sel), 0, (int)((size_t)N * CP * PWb), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(gx, 0, (int)((size_t)N * CIN * Pb), 0x00020000);
    const uint32_t pb = (uint32_t)p * 4u;
    const uint32_t g_off = valid ? (uint32_t)lk * Pb + pb : kOut;
    const uint32_t x_off = valid ? 4u * (uint32_t)lk * Pb + pb : kOut;
    uint32_t s_off, s_soff;
    if (SEL == 2) { s_off = valid ? (uint32_t)li * 4u : kOut; s_soff = (uint32_t)(n * PW + tile) * 128u; }
    else { s_off = valid ? (uint32_t)lk * PWb + (uint32_t)tile * 4u : kOut; s_soff = (uint32_t)n * CP * PWb; }
    float g[16]; uint32_t sw[16];
    bwd_issue<SEL, AUX>(gr, sr, g_off, (uint32_t)n * CP * Pb, s_off, s_soff, Pb, PWb, g, sw);
    const float none[32] = {0};
    bwd_compute<SEL, AUX, false>(g, sw, w_s, none, w_s + 2 * CP * 33, xr, x_off, (uint32_t)n * CIN * Pb, Pb, li, lk);
}

template <int D, int AUX, int G, bool HOISTW>
__global__ __launch_bounds__(256, G) void k_bwd_persist(const float *__restrict__ gy, const uint32_t *__restrict__ sel,
                                                     const float *__restrict__ weight, const float *__restrict__ gscale,
                                                     float *__restrict__ gx, int N, int P, int PW, int iters) {
    extern __shared__ float w_s[];
    stage_bwd(weight, gscale, w_s);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const int nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
    float wr[32];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        wr[s] = HOISTW ? w_s[(2 * s + lk) * 33 + li] : 0.0f;
        wr[16 + s] = HOISTW ? w_s[(CP + 2 * s + lk) * 33 + li] : 0.0f;
        if (HOISTW) { asm volatile("" : "+v"(wr[s])); asm volatile("" : "+v"(wr[16 + s])); }
    }
    const uint32_t Pb = (uint32_t)P * 4u, PWb = (uint32_t)PW * 4u;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gy), 0, (int)((size_t)N * CP * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(sel), 0, (int)((size_t)N * CP * PWb), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(gx, 0, (int)((size_t)N * CIN * Pb), 0x00020000);
    const int dn = nwaves / PW, dt = nwaves - dn * PW;
    int ln = __builtin_amdgcn_readfirstlane(w0 / PW), lt = __builtin_amdgcn_readfirstlane(w0 % PW);
    int cn = ln, ct = lt;
    auto advance = [&](int &n, int &t) {
        n += dn; t += dt;
        if (t >= PW) { t -= PW; ++n; }
    };
    auto issue = [&](float (&g)[16], uint32_t (&sw)[16]) {
        const int p = lt * 32 + li;
        const bool valid = p < P && ln < N;
        const uint32_t g_off = valid ? (uint32_t)lk * Pb + (uint32_t)p * 4u : kOut;
        const uint32_t s_off = valid ? (uint32_t)li * 4u : kOut;
        bwd_issue<2, AUX>(gr, sr, g_off, (uint32_t)ln * CP * Pb, s_off, (uint32_t)(ln * PW + lt) * 128u, Pb, PWb, g, sw);
        advance(ln, lt);
    };
    auto compute = [&](const float (&g)[16], const uint32_t (&sw)[16]) {
        const int p = ct * 32 + li;
        const bool valid = p < P && cn < N;
        const uint32_t x_off = valid ? 4u * (uint32_t)lk * Pb + (uint32_t)p * 4u : kOut;
        bwd_compute<2, AUX, HOISTW>(g, sw, w_s + opaque_zero(), wr, w_s + 2 * CP * 33 + opaque_zero(), xr, x_off, (uint32_t)cn * CIN * Pb, Pb, li, lk);
        advance(cn, ct);
    };
    float g[D][16]; uint32_t sw[D][16];
#pragma unroll
    for (int j = 0; j < D - 1; ++j) issue(g[j], sw[j]);
#pragma unroll 1
    for (int k = 0; k < iters; k += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            issue(g[(j + D - 1) % D], sw[(j + D - 1) % D]);
            compute(g[j], sw[j]);
        }
    }
}

__global__ void fill(float *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xffff) * (1.0f / 32768.0f) - 1.0f;
    }
}

int main(int argc, char **argv) {
    const int N = 128, P = argc > 1 ? atoi(argv[1]) : 8080, PW = (P + 31) / 32;
    const size_t xe = (size_t)N * CIN * P;
    float *x, *y, *yref, *w, *par; uint32_t *sel;
    CK(hipMalloc(&x, xe * 4)); CK(hipMalloc(&y, xe * 4)); CK(hipMalloc(&yref, xe * 4)); CK(hipMalloc(&w, 2 * CP * CIN * 4));
    CK(hipMalloc(&par, 128 * 4));
    const size_t selb = (size_t)N * PW * 64 * 4;
    CK(hipMalloc(&sel, selb));
    fill<<<1024, 256>>>(x, xe, 1u); fill<<<8, 256>>>(w, 2 * CP * CIN, 2u);
    {
        std::vector<float> h(128);
        for (int i = 0; i < 128; ++i) h[i] = i < 96 ? 0.01f * (i % 7) : 1.0f + 0.01f * (i % 5);
        CK(hipMemcpy(par, h.data(), 512, hipMemcpyHostToDevice));
    }
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("P = %d, CUs = %d\n", P, cus);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (2 * 32 * 33 + 128) * 4;
    std::vector<float> href(xe), hy(xe);
    bool have_ref = false;
    auto check = [&](const char *name) {
        CK(hipMemcpy(hy.data(), y, xe * 4, hipMemcpyDeviceToHost));
        if (!have_ref) { href = hy; have_ref = true; return; }
        size_t bad = 0;
        for (size_t i = 0; i < xe; ++i) bad += memcmp(&hy[i], &href[i], 4) != 0;
        if (bad) printf("   !! %s: %zu of %zu outputs differ from the first variant\n", name, bad, xe);
    };
    auto timeit = [&](const char *name, auto launch) {
        CK(hipMemset(y, 0xff, xe * 4));
        launch(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
        check(name);
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 3; ++rep) {
            float ms;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 30; ++i) launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%-58s %7.1f us (best of 3 x 30; mean %6.1f)\n", name, 1e3 * best / 30, 1e3 * sum / 90);
    };
    const dim3 gb((PW + 3) / 4, N);
#define BASE(SEL, AUX) [&] { hipLaunchKernelGGL((k_base<SEL, AUX>), gb, dim3(256), lds, 0, x, w, par, y, sel, N, P, PW); }
#define PERS(D, SEL, AUX, G, H) [&] { const int wv = cus * G * 4, it = ((N * PW + wv - 1) / wv + D - 1) / D * D; \
        hipLaunchKernelGGL((k_persist<D, SEL, AUX, G, H>), dim3(cus * G), dim3(256), lds, 0, x, w, par, y, sel, N, P, PW, it); }
    printf("== forward\n");
    timeit("shipped structure (sel words)", BASE(0, 0));
    timeit("  no selection stores (timing only)", BASE(1, 0));
    timeit("  lane-mask selection, one store per wave", BASE(2, 0));
    timeit("  lane-mask, nt loads + stores", BASE(2, 2));
    timeit("persistent D=2 G=2 W in registers", PERS(2, 2, 0, 2, true));
    timeit("persistent D=4 G=2 W in registers", PERS(4, 2, 0, 2, true));
    timeit("persistent D=2 G=3 W in registers", PERS(2, 2, 0, 3, true));
    timeit("persistent D=3 G=3 W in registers", PERS(3, 2, 0, 3, true));
    timeit("persistent D=4 G=3 W in registers", PERS(4, 2, 0, 3, true));
    timeit("persistent D=2 G=4 W in registers", PERS(2, 2, 0, 4, true));
    timeit("persistent D=3 G=4 W in registers", PERS(3, 2, 0, 4, true));
    timeit("persistent D=3 G=4 W from LDS", PERS(3, 2, 0, 4, false));
    timeit("persistent D=4 G=4 W from LDS", PERS(4, 2, 0, 4, false));
    timeit("persistent D=2 G=5 W from LDS", PERS(2, 2, 0, 5, false));
    timeit("persistent D=3 G=5 W from LDS", PERS(3, 2, 0, 5, false));
    timeit("persistent D=2 G=6 W from LDS", PERS(2, 2, 0, 6, false));
    timeit("persistent D=3 G=3 W in registers, sel words", PERS(3, 0, 0, 3, true));
    timeit("persistent D=3 G=3 W in registers, no sel (timing only)", PERS(3, 1, 0, 3, true));
    timeit("persistent D=3 G=3 W in registers, nt", PERS(3, 2, 2, 3, true));
    timeit("persistent D=2 G=4 W in registers, nt", PERS(2, 2, 2, 4, true));

    printf("== backward (gy = the forward's y, selection from the forward)\n");
    float *gx = x;  // overwritten: the forward runs are done
    have_ref = false;
    float *gy = yref;
    fill<<<1024, 256>>>(gy, xe, 3u);
    // selection bits in both formats from one forward run each
    uint32_t *sel0; CK(hipMalloc(&sel0, selb));
    float *xin; CK(hipMalloc(&xin, xe * 4)); fill<<<1024, 256>>>(xin, xe, 1u);
    hipLaunchKernelGGL((k_base<0, 0>), gb, dim3(256), lds, 0, xin, w, par, y, sel0, N, P, PW);
    hipLaunchKernelGGL((k_base<2, 0>), gb, dim3(256), lds, 0, xin, w, par, y, sel, N, P, PW);
    CK(hipDeviceSynchronize());
    float *ysave = y; y = gx;   // check() reads y: point it at gx
#define BBASE(SEL, AUX, S) [&] { hipLaunchKernelGGL((k_bwd_base<SEL, AUX>), gb, dim3(256), lds, 0, gy, S, w, par + 96, gx, N, P, PW); }
#define BPERS(D, AUX, G, H) [&] { const int wv = cus * G * 4, it = ((N * PW + wv - 1) / wv + D - 1) / D * D; \
        hipLaunchKernelGGL((k_bwd_persist<D, AUX, G, H>), dim3(cus * G), dim3(256), lds, 0, gy, sel, w, par + 96, gx, N, P, PW, it); }
    timeit("shipped structure (sel words, 16 broadcast loads)", BBASE(0, 0, sel0));
    timeit("  lane masks, one selection load", BBASE(2, 0, sel));
    timeit("  lane masks, nt", BBASE(2, 2, sel));
    timeit("persistent D=2 G=2 W in registers", BPERS(2, 0, 2, true));
    timeit("persistent D=4 G=2 W in registers", BPERS(4, 0, 2, true));
    timeit("persistent D=2 G=3 W in registers", BPERS(2, 0, 3, true));
    timeit("persistent D=3 G=3 W in registers", BPERS(3, 0, 3, true));
    timeit("persistent D=4 G=3 W in registers", BPERS(4, 0, 3, true));
    timeit("persistent D=2 G=4 W in registers", BPERS(2, 0, 4, true));
    timeit("persistent D=3 G=4 W in registers", BPERS(3, 0, 4, true));
    timeit("persistent D=4 G=4 W from LDS", BPERS(4, 0, 4, false));
    timeit("persistent D=2 G=5 W from LDS", BPERS(2, 0, 5, false));
    timeit("persistent D=3 G=5 W from LDS", BPERS(3, 0, 5, false));
    timeit("persistent D=2 G=6 W from LDS", BPERS(2, 0, 6, false));
    timeit("persistent D=3 G=3 W in registers, nt", BPERS(3, 2, 3, true));
    y = ysave;
    return 0;
}
