// Harness (round 5): conv1x1_mfm_forward at L3's shape (N = 128, Cin = 32, C = 32, P = 8080) - one 32-pixel tile per wave (the
// shipped structure) against a wave that walks TPW tiles with the next tile's fragments requested before this tile's matrix
// instructions.  hipcc --offload-arch=gfx950 -O3 tools/tune_conv1x1_pipe.hip -o tools/_tune_c11p.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bool takes_b(float a, float b) { return !(a != a) && !(a >= b); }

template <int TPW, bool PIPE>
__global__ __launch_bounds__(256) void k(const float *__restrict__ x, const float *__restrict__ weight,
                                         const float *__restrict__ bias, float *__restrict__ y, uint32_t *__restrict__ sel,
                                         int C, long P, long PW) {
    constexpr int CIN = 32, CP = 32, PITCH = CIN + 1;
    extern __shared__ float w_s[];
    float *par = w_s + 2 * CP * PITCH;
    {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = weight[threadIdx.x + j * 256];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = threadIdx.x + j * 256, row = i / CIN, ci = i - row * CIN;
            w_s[row * PITCH + ci] = wv[j];
        }
        if (threadIdx.x < 64) par[threadIdx.x] = bias[threadIdx.x];
    }
    __syncthreads();
    const long n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const uint32_t Pb = (uint32_t)P * 4u, PWb = (uint32_t)PW * 4u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + n * CIN * P), 0, (int)(CIN * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + n * (long)C * P, 0, (int)(C * Pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(sel + n * (long)C * PW, 0, (int)(C * PWb), 0x00020000);
    constexpr uint32_t kOut = 0x80000000u;
    const float *wa = w_s + li * PITCH + lk;
    const float *wb = wa + CP * PITCH;
    auto offs = [&](int i, uint32_t &x_off, uint32_t &y_off, uint32_t &s_off, bool &valid) {
        const long p0 = (((long)blockIdx.x * TPW + i) * 4 + wave) * 32;
        const long p = p0 + li;
        valid = p < P;
        const uint32_t pb = (uint32_t)p * 4u;
        x_off = valid ? (uint32_t)lk * Pb + pb : kOut;
        y_off = valid ? 4u * (uint32_t)lk * Pb + pb : kOut;
        s_off = (li == 0 && valid) ? 4u * (uint32_t)lk * PWb + (uint32_t)(p >> 5) * 4u : kOut;
    };
    auto load = [&](uint32_t x_off, float (&xb)[CIN / 2]) {
#pragma unroll
        for (int s = 0; s < CIN / 2; ++s)
            xb[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, x_off, (uint32_t)(2 * s) * Pb, 0));
    };
    auto compute = [&](const float (&xb)[CIN / 2], uint32_t y_off, uint32_t s_off, bool valid) {
        f32x16 acc_a = {0}, acc_b = {0};
#pragma unroll
        for (int s = 0; s < CIN / 2; ++s) {
            acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * s], xb[s], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[2 * s], xb[s], acc_b, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c_lo = (r & 3) + 8 * (r >> 2), c = c_lo + 4 * lk;
            const float va = acc_a[r] + par[c], vb = acc_b[r] + par[CP + c];
            const bool tb = takes_b(va, vb);
            const unsigned long long word = __ballot(valid && tb);
            const float v = tb ? vb : va;
            const uint32_t sw = lk ? (uint32_t)(word >> 32) : (uint32_t)word;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yr, y_off, (uint32_t)c_lo * Pb, 0);
            __builtin_amdgcn_raw_buffer_store_b32(sw, sr, s_off, (uint32_t)c_lo * PWb, 0);
        }
    };
    if (!PIPE) {
        for (int i = 0; i < TPW; ++i) {
            uint32_t xo, yo, so; bool valid;
            offs(i, xo, yo, so, valid);
            float xb[CIN / 2];
            load(xo, xb);
            compute(xb, yo, so, valid);
        }
    } else {
        static_assert(!PIPE || TPW % 2 == 0, "pairs");
        float xa[CIN / 2], xb[CIN / 2];
        uint32_t xo, yo0, so0, yo1, so1; bool v0, v1;
        offs(0, xo, yo0, so0, v0);
        load(xo, xa);
#pragma unroll 1
        for (int i = 0; i < TPW; i += 2) {
            offs(i + 1, xo, yo1, so1, v1);
            load(xo, xb);
            compute(xa, yo0, so0, v0);
            if (i + 2 < TPW) {
                offs(i + 2, xo, yo0, so0, v0);
                load(xo, xa);
            }
            compute(xb, yo1, so1, v1);
        }
    }
}

int main() {
    const long N = 128, P = 8080; const int C = 32, CIN = 32;
    float *x, *y, *w, *b; uint32_t *sel;
    CK(hipMalloc(&x, N * CIN * P * 4)); CK(hipMalloc(&y, N * C * P * 4)); CK(hipMalloc(&w, 2 * C * CIN * 4)); CK(hipMalloc(&b, 2 * C * 4));
    CK(hipMalloc(&sel, N * C * ((P + 31) / 32) * 4));
    CK(hipMemset(x, 0, N * CIN * P * 4)); CK(hipMemset(w, 0, 2 * C * CIN * 4)); CK(hipMemset(b, 0, 2 * C * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (2 * 32 * 33 + 64) * 4;
    for (int rep = 0; rep < 2; ++rep) {
        auto run = [&](const char *name, auto kern, int tpw) {
            float ms;
            const dim3 grid((P + 128 * tpw - 1) / (128 * tpw), N);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, x, w, b, y, sel, C, P, (P + 31) / 32);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, x, w, b, y, sel, C, P, (P + 31) / 32);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-40s %7.1f us\n", name, 1e3 * ms / 50);
        };
        run("one tile per wave", k<1, false>, 1);
        run("2 tiles per wave, sequential", k<2, false>, 2);
        run("4 tiles per wave, sequential", k<4, false>, 4);
        run("2 tiles per wave, pipelined", k<2, true>, 2);
        run("4 tiles per wave, pipelined", k<4, true>, 4);
        run("8 tiles per wave, pipelined", k<8, true>, 8);
        run("16 tiles per wave, pipelined", k<16, true>, 16);
        printf("--\n");
    }
    return 0;
}
