// Ablation harness for conv1x1_mfm_forward_kernel<32, 1> at L3's shape (N = 128, Cin = 32, C = 32, P = 8080):
// which part of the kernel costs the time?  Variants switch off the MFMA chain, the LDS weight staging, the epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bool takes_b(float a, float b) { return !(a != a) && !(a >= b); }
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <bool MFMA, bool STAGE, bool EPI, bool XLOAD>
__global__ __launch_bounds__(256) void k(const float *__restrict__ x, const float *__restrict__ weight,
                                         const float *__restrict__ bias, float *__restrict__ y, uint32_t *__restrict__ sel,
                                         int C, long P, long PW) {
    constexpr int CIN = 32, CP = 32, PITCH = CIN + 1;
    extern __shared__ float w_s[];
    if (STAGE) {
        for (int i = threadIdx.x; i < 2 * CP * CIN; i += 256) {
            const int row = i / CIN, ci = i - row * CIN;
            const int half = row / CP, c = row - half * CP;
            w_s[row * PITCH + ci] = c < C ? weight[(long)(half * C + c) * CIN + ci] : 0.0f;
        }
        __syncthreads();
    }
    const long n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const long p = (long)blockIdx.x * 128 + wave * 32 + li;
    const bool valid = p < P;
    if ((long)blockIdx.x * 128 + wave * 32 >= P) return;
    float xb[CIN / 2];
    const float *xn = x + n * CIN * P + (valid ? p : 0);
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) xb[s] = (XLOAD && valid) ? xn[(long)(2 * s + lk) * P] : 1.0f;
    float *yn = y + n * (long)C * P + p;
    uint32_t *sn = sel + n * (long)C * PW + (p >> 5);
    f32x16 acc_a = {0}, acc_b = {0};
    const float *wa = w_s + li * PITCH + lk;
    const float *wb = wa + CP * PITCH;
    if (MFMA) {
#pragma unroll
        for (int s = 0; s < CIN / 2; ++s) {
            acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(STAGE ? wa[2 * s] : 0.5f, xb[s], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(STAGE ? wb[2 * s] : 0.25f, xb[s], acc_b, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int s = 0; s < CIN / 2; ++s) { acc_a[s] = xb[s]; acc_b[s] = xb[s] * 0.5f; }
    }
    if (EPI) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = mfma_row(r, lane);
            const bool live = c < C;
            float va = acc_a[r], vb = acc_b[r];
            if (bias && live) { va += bias[c]; vb += bias[c + C]; }
            const bool tb = live && takes_b(va, vb);
            const unsigned long long word = __ballot(valid && tb);
            float v = tb ? vb : va;
            if (live && valid) yn[(long)c * P] = v;
            if (live && li == 0) sn[(long)c * PW] = lk ? (uint32_t)(word >> 32) : (uint32_t)word;
        }
    } else {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc_a[r] + acc_b[r];
        if (valid) yn[0] = t;
    }
}

int main() {
    const long N = 128, P = 8080; const int C = 32, CIN = 32;
    float *x, *y, *w, *b; uint32_t *sel;
    CK(hipMalloc(&x, N * CIN * P * 4)); CK(hipMalloc(&y, N * C * P * 4)); CK(hipMalloc(&w, 2 * C * CIN * 4)); CK(hipMalloc(&b, 2 * C * 4));
    CK(hipMalloc(&sel, N * C * ((P + 31) / 32) * 4));
    CK(hipMemset(x, 0, N * CIN * P * 4)); CK(hipMemset(w, 0, 2 * C * CIN * 4)); CK(hipMemset(b, 0, 2 * C * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 2 * 32 * 33 * 4;
    const dim3 grid((P + 127) / 128, N);
    for (int rep = 0; rep < 2; ++rep) {
        auto run = [&](const char *name, auto kern) {
            float ms;
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, x, w, b, y, sel, C, P, (P + 31) / 32);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, x, w, b, y, sel, C, P, (P + 31) / 32);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-40s %7.1f us\n", name, 1e3 * ms / 50);
        };
        run("full", k<true, true, true, true>);
        run("no MFMA", k<false, true, true, true>);
        run("no LDS staging", k<true, false, true, true>);
        run("no epilogue (1 store)", k<true, true, false, true>);
        run("no x loads", k<true, true, true, false>);
        run("no MFMA, no staging", k<false, false, true, true>);
        run("only loads + full epilogue, no stage/mfma", k<false, false, true, true>);
        run("MFMA + staging only (no x, no epi)", k<true, true, false, false>);
        printf("--\n");
    }
    return 0;
}
