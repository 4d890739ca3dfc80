// Does the width of the contiguous segment a half-wave touches matter for NCHW "one pixel column per lane" kernels?
// Reads CIN channel rows and writes C channel rows of a (N, C, P) tensor with (A) 32 pixels x 4 B per half-wave
// (the MFMA conv1x1 kernel's pattern) and (B) 32 lanes x 16 B.  No arithmetic to speak of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void rows_dword(const float *__restrict__ x, float *__restrict__ y, long P) {
    const long n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const long p = (long)blockIdx.x * 128 + wave * 32 + li;
    if (p >= P) return;
    float v[CIN / 2];
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) v[s] = x[(n * CIN + 2 * s + lk) * P + p];
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) acc += v[s];
#pragma unroll
    for (int r = 0; r < COUT / 2; ++r) y[(n * COUT + 2 * r + lk) * P + p] = acc + r;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void rows_dwordx4(const float *__restrict__ x, float *__restrict__ y, long P) {
    const long n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const long p = (long)blockIdx.x * 512 + wave * 128 + li * 4;
    if (p >= P) return;
    float4 v[CIN / 2];
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) v[s] = *reinterpret_cast<const float4 *>(&x[(n * CIN + 2 * s + lk) * P + p]);
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) { acc.x += v[s].x; acc.y += v[s].y; acc.z += v[s].z; acc.w += v[s].w; }
#pragma unroll
    for (int r = 0; r < COUT / 2; ++r)
        *reinterpret_cast<float4 *>(&y[(n * COUT + 2 * r + lk) * P + p]) = make_float4(acc.x + r, acc.y, acc.z, acc.w);
}

// thread = pixel, loops over all channels (the VALU kernel's pattern): 64 lanes x 4 B contiguous
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void rows_thread_per_pixel(const float *__restrict__ x, float *__restrict__ y, long P) {
    const long n = blockIdx.y;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float v[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) v[c] = x[(n * CIN + c) * P + p];
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc += v[c];
#pragma unroll
    for (int c = 0; c < COUT; ++c) y[(n * COUT + c) * P + p] = acc + c;
}

int main() {
    const long N = 128, P = 8080;
    constexpr int CIN = 32, COUT = 32;
    float *x, *y;
    CK(hipMalloc(&x, N * CIN * P * 4)); CK(hipMalloc(&y, N * COUT * P * 4));
    CK(hipMemset(x, 0, N * CIN * P * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 4.0 * N * P * (CIN + COUT);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        auto run = [&](const char *name, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 50; ++i) launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-34s %7.1f us %6.0f GB/s\n", name, 1e3 * ms / 50, bytes / (ms / 50 * 1e-3) / 1e9);
        };
        run("half-wave = 32 px x 4 B (MFMA map)", [&] { hipLaunchKernelGGL((rows_dword<CIN, COUT>), dim3((P + 127) / 128, N), dim3(256), 0, 0, x, y, P); });
        run("half-wave = 32 lanes x 16 B", [&] { hipLaunchKernelGGL((rows_dwordx4<CIN, COUT>), dim3((P + 511) / 512, N), dim3(256), 0, 0, x, y, P); });
        run("thread = pixel, 64 lanes x 4 B", [&] { hipLaunchKernelGGL((rows_thread_per_pixel<CIN, COUT>), dim3((P + 255) / 256, N), dim3(256), 0, 0, x, y, P); });
    }
    return 0;
}
